"""SafeCemMpc, reference simba/policies/safe_cem_mpc.py:7-120: CemMpc whose objective masks done trajectories
before the reward and subtracts 100 from candidates a per-step Beta posterior over particle cost counts calls
unsafe (:76-96,110-120).  Constructor kwargs as :8-19, including the YAML spelling ``posterior_mean_threashold``.
``optimize_for_safety`` / ``compute_mean_costs`` (:40-74,98-108) have no callers in the reference and are not
provided."""
from .cem_mpc import CemMpc


class SafeCemMpc(CemMpc):
    variant = 'safe'

    def __init__(self, model, environment, horizon, iterations, smoothing, n_samples, n_elite, particles,
                 stddev_threshold, noise_stddev, posterior_mean_threashold, **kwargs):
        super().__init__(model, environment, horizon, iterations, smoothing, n_samples, n_elite, particles,
                         stddev_threshold, noise_stddev, **kwargs)
        self.cost = getattr(environment, 'get_cost', None)
        self.posterior_mean_threashold = posterior_mean_threashold

    def _extra_config(self):
        return dict(posterior_mean_threashold=self.posterior_mean_threashold)

    def _objective_extra_config(self):
        return dict(posterior_mean_threashold=self.posterior_mean_threashold)
