"""Golden vector for the BBS bending matrix, generated from the reference's own bending_ur (bbs.cc compiled by
oracle/Makefile into oracle/_ref/libbbs_ref.so).  Run in the build container: python tests/golden/make_golden_sfn.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle

bbs = (-0.6, 0.62, 13, -0.45, 0.5, 15, 1)
lam = 0.7
B = oracle.ref_bbs_bending(bbs, lam)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "bending_13x15.npz"), bbs=np.array(bbs[:6], float), lam=lam, bending=B)
print("bending_13x15.npz", B.shape, np.abs(B).max())
