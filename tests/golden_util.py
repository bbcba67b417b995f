"""Loading of the committed fixtures under tests/golden/ (made by tests/golden/make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NS_FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ns_*.npz")))
LS_FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ls_*.npz")))
PRM_FIELDS = ["physical_type", "linearization", "beta", "tau_grad_div", "density", "viscosity", "damping",
              "density_diff", "weight", "weight_old", "weight_old_old", "tau1", "extrap_old", "extrap_old_old"]


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def prm_dict(d):
    p = dict(zip(PRM_FIELDS, d["prm"]))
    p["physical_type"], p["linearization"] = int(p["physical_type"]), int(p["linearization"])
    return p


class FixedTimeStepping:
    """TimeStepping stand-in that returns the scalars stored in a fixture"""

    def __init__(self, p):
        self._p = p
        self.factor_extrapol_old, self.factor_extrapol_old_old = p["extrap_old"], p["extrap_old_old"]

    def weight(self):
        return self._p["weight"]

    def weight_old(self):
        return self._p["weight_old"]

    def weight_old_old(self):
        return self._p["weight_old_old"]

    def tau1(self):
        return self._p["tau1"]

    def tau2(self):
        return 0.0
