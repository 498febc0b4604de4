"""Sliding-window batch assembly: oracle/data_oracle.py against the outputs of the reference's own
`__getitem__` (tests/golden/make_golden_batch.py).  CPU only; bit-exact."""
import glob
import os

import numpy as np
import pytest

from oracle import data_oracle as D
from tests.helpers import GOLDEN
from tests.golden.batch_inputs import make_tables

CASES = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "batch_*.npz")))


@pytest.mark.parametrize("case", CASES)
def test_getitem_oracle_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, case))
    tb = make_tables(int(g["seed"]), str(g["modality"]))
    assert int(g["n"]) == len(tb["windows"])
    for i in range(int(g["n"])):
        v, a, t, label, meta = D.getitem(tb, i, g["va%d" % i], g["aa%d" % i])
        assert np.array_equal(v, g["v%d" % i]) and np.array_equal(a, g["a%d" % i])
        assert np.array_equal(t, g["t%d" % i])
        for k in ("verb", "noun", "action", "class_id"):
            assert np.array_equal(label[k], g["%s%d" % (k, i)]), k
        assert np.array_equal(meta["v_action_ids"], g["vid%d" % i]) and np.array_equal(meta["a_action_ids"], g["aid%d" % i])
