"""Model registry with the reference's interface (bayes_drt/stan_models.py): `load_pickle(path)` returns the
object `Inverter._get_stan_model` hands to `fit`.  Here the "pickle" is a GPU-backed engine.StanModel; nothing is
compiled at import time (the reference compiles 14 Stan programs for ~20 min on first import, README.md:37)."""
import os
import pickle

from .engine import MODEL_NAMES, UNSUPPORTED, StanModel

model_dict = {name: name.replace('_StanModel.pkl', '_modelcode.txt') for name in MODEL_NAMES}


def load_pickle(file):
    name = os.path.basename(file)
    if name in model_dict or name in UNSUPPORTED:
        return StanModel(name)
    with open(file, 'rb') as f:
        return pickle.load(f)


def save_pickle(obj, file):
    with open(file, 'wb') as f:
        pickle.dump(obj, f, pickle.HIGHEST_PROTOCOL)
    print('Dumped pickle to {}'.format(file))
