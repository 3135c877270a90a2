"""reference simba/infrastructure/common.py:5-6."""


def standardize_name(name):
    return ''.join(w.capitalize() for w in name.split('_'))
