"""Helpers to read tests/golden/model_*.npz (written by oracle/make_golden.py)."""
import ast
import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def model_cases():
    return sorted(os.path.basename(p)[len("model_"):-len(".npz")]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "model_*.npz")))


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, f"model_{name}.npz"), allow_pickle=False)
        self.z = z
        self.meta = ast.literal_eval(str(z["meta"]))
        self.batch = self.meta.pop("batch")
        self.T = self.meta.pop("T")
        self.cond = self.meta.pop("cond")
        self.cfg = self.meta
        self.lo, self.hi = (int(v) for v in z["loss_window"])

    def t(self, key):
        return torch.from_numpy(np.array(self.z[key]))

    def group(self, prefix):
        return {k[len(prefix):]: torch.from_numpy(np.array(self.z[k])) for k in self.z.files if k.startswith(prefix)}

    def state_dict(self):
        return self.group("sd/")

    def cond_kwargs(self):
        if self.cond == "ids":
            return {"odor_ids": self.t("odor_ids")}
        if self.cond == "emb":
            return {"cond_emb": self.t("cond_emb")}
        return {}
