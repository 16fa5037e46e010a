"""Shared test helpers (oracle loading, golden fixtures).  Test-only."""
import ctypes
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

KIND_FILES = {
    ('transe', 2): 'ref_transe.npz', ('transe', 1): 'ref_transe_l1.npz',
    ('transh', 2): 'ref_transh.npz', ('transd', 2): 'ref_transd.npz',
    ('distmult', 2): 'ref_distmult.npz', ('complex', 2): 'ref_complex.npz',
}
N_TABLES = {'transe': 2, 'transh': 3, 'transd': 4, 'distmult': 2, 'complex': 4}


def load_golden(kind, p=2):
    z = np.load(os.path.join(GOLDEN, KIND_FILES[(kind, p)]))
    tables = [torch.from_numpy(z['table%d' % i]) for i in range(N_TABLES[kind])]
    return z, tables


def oracle_clib():
    """Build (if needed) and load the C oracle."""
    so = os.path.join(ROOT, 'oracle', '_build', 'libkge_oracle.so')
    src = os.path.join(ROOT, 'oracle', 'kge_oracle.c')
    if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    return ctypes.CDLL(so)


def fptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def dict_to_csr(dictionary, key1, key2):
    """Per-query CSR view of dictionary[(key1_i, key2_i)] (python, test-only)."""
    has, off, tgt = [], [0], []
    for a, b in zip(np.asarray(key1).tolist(), np.asarray(key2).tolist()):
        if (a, b) in dictionary:
            has.append(1)
            tgt.extend(sorted(dictionary[(a, b)]))
        else:
            has.append(0)
        off.append(len(tgt))
    return (np.array(has, dtype=np.uint8), np.array(off, dtype=np.int64),
            np.array(tgt, dtype=np.int64))
