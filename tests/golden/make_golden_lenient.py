#!/usr/bin/env python3
"""Golden vectors for inputs pandas reads and a strict parser of the format (README.md:127-134) would refuse: what the REAL
reference (/root/reference, build container only) returns for them -- helpers.readStates (helpers.py:152-155) for the states,
the locationArr of scores.py:161 printed through writeScores' template (scores.py:526-531) for the first three columns.

    python tests/golden/make_golden_lenient.py       # rewrites tests/golden/lenient.npz

The input texts are authored here (they are data, not reference source): a blank line inside the file, blanks around numbers,
an explicit plus sign, a float-formatted state, all of them together, and -- for contrast -- a value pandas cannot read either.
Stored per case: the file's bytes, the reference's int array (0-based states as readStates returns them) and the three location
columns as the reference prints them."""
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden import import_reference  # noqa: E402

HERE = Path(__file__).resolve().parent

CASES = {
    "blank_line": "chr1\t0\t200\t1\t2\t3\nchr1\t200\t400\t4\t5\t6\n\nchr1\t400\t600\t7\t8\t9\nchr1\t600\t800\t18\t1\t2\n",
    "blanks_around": "chr1\t0\t200\t 1\t2 \t 3 \nchr1\t200\t400\t4\t 5\t6\n",
    "plus_sign": "chr1\t0\t200\t+1\t2\t+3\nchr1\t200\t400\t4\t+5\t6\n",
    "float_state": "chr1\t0\t200\t1.0\t2\t3\nchr1\t200\t400\t4\t5.0\t6\n",
    "all_together": "chr1\t0\t200\t 1\t+2\t3.0\n\nchr1\t 200\t400 \t4\t5\t 6 \n",
    "crlf": "chr1\t0\t200\t1\t2\t3\r\nchr1\t200\t400\t4\t5\t6\r\n",
}


def main():
    _ex, _ec, _sc, hp = import_reference()
    import pandas as pd
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, text in CASES.items():
            p = Path(td) / (name + ".txt")
            p.write_bytes(text.encode())
            nrows = hp.countRows(p)                                              # what the reference would split into row ranges
            states = hp.readStates(file1Path=p, rowsToCalc=(0, nrows), verbose=False)
            loc = pd.read_table(p, header=None, sep="\t", usecols=[0, 1, 2]).to_numpy()     # scores.py:161
            out[name + "_text"] = np.frombuffer(text.encode(), dtype=np.uint8)
            out[name + "_states"] = np.asarray(states, dtype=np.int64)
            out[name + "_loc"] = np.array(["{0[0]}\t{0[1]}\t{0[2]}".format(r) for r in loc])   # scores.py:526-531's prefix
            out[name + "_countRows"] = np.int64(nrows)
            print(name, "countRows", nrows, "states", states.shape, states.tolist(), list(out[name + "_loc"]))
    np.savez_compressed(HERE / "lenient.npz", **out)


if __name__ == "__main__":
    main()
