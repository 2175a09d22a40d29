"""Helpers shared by the tests: golden-fixture loading."""
import os

import numpy as np
from scipy.sparse import csc_matrix

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def unpack(d):
    """(AD, DP) int64 CSC, as the reference's read_cellSNP() returns them."""
    shape = tuple(int(x) for x in d["shape"])
    AD = csc_matrix((d["AD_data"], d["AD_indices"], d["AD_indptr"]), shape=shape)
    DP = csc_matrix((d["DP_data"], d["DP_indices"], d["DP_indptr"]), shape=shape)
    return AD, DP


def c1():
    return unpack(load("c1_data"))


def mito():
    return unpack(load("mito_data"))
