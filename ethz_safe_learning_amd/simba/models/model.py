"""reference simba/models/model.py:1-20."""


class BaseModel(object):
    def __init__(self, inputs_dim, outputs_dim):
        self.inputs_dim = inputs_dim
        self.outputs_dim = outputs_dim

    def build(self):
        raise NotImplementedError

    def fit(self, inputs, targets):
        raise NotImplementedError

    def predict(self, inputs):
        raise NotImplementedError

    def save(self):
        raise NotImplementedError

    def load(self):
        raise NotImplementedError
