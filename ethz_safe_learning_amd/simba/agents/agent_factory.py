"""make_agent(config, environment): wire agent / policy / model sections of the merged config together
(reference simba/agents/agent_factory.py:5-24; ``train_epochs`` of the model is the experiment's ``train_iterations``)."""
from ..infrastructure.common import standardize_name
from . import mbrl_agent

_AGENTS = dict(MbrlAgent=mbrl_agent.MbrlAgent)


def make_agent(config, environment):
    name = config['options']['agent']
    assert name in config['agents'], "Specified agent does not exist."
    agent_cls = _AGENTS[standardize_name(name)]
    agent_params = config['agents'][name]
    assert agent_params['policy'] in config['policies'], "Specified policy does not exist."
    assert agent_params['model'] in config['models'], "Specified model does not exist."
    assert len(environment.action_space.shape) == 1 and len(environment.observation_space.shape) == 1, \
        "No support for non-flat action/observation spaces."
    model_params = dict(config['models'][agent_params['model']], train_epochs=config['options']['train_iterations'])
    kwargs = dict(agent_params, **config['agents']['agent'])
    kwargs.update(policy_params=dict(config['policies'][agent_params['policy']]), model_params=model_params)
    return agent_cls(environment=environment, **kwargs)
