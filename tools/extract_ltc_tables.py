#!/usr/bin/env python3
"""Transcribe the LTC fit tables NUMERICALLY into a binary fixture.

Reads the numeric initialisers of tabM / tabAmplitude in the reference's
src/LTC/ltc_ggx.cpp and src/LTC/ltc_beckmann.cpp as text (data, not code) and
writes rgk_amd/data/ltc_{ggx,beckmann}.f32: 64*64 records of float32
{m0, m2, m4, m6, amplitude}.  Only those four matrix entries vary (m8 == 1, the rest
are 0 -- asserted here); the doubles are rounded to float32 exactly as the
reference's `operator glm::mat3()` does on every read (src/LTC/ltc.hpp:6-9).
tabMinv is unused by the reference and is not transcribed.

Run in the build container only (needs /root/reference):
    python tools/extract_ltc_tables.py
"""
import os
import re
import sys

import numpy as np

REF = os.environ.get("RGK_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rgk_amd", "data")


def extract(name):
    src = open(os.path.join(REF, "src", "LTC", f"ltc_{name}.cpp")).read()
    m = re.search(r"const mat33 tabM\[size\*size\] = \{(.*?)\n\};", src, re.S)
    rows = re.findall(r"\{([^{}]*)\}", m.group(1))
    M = np.array([[float(x) for x in r.split(",")] for r in rows], dtype=np.float64)
    assert M.shape == (4096, 9), M.shape
    assert np.all(M[:, [1, 3, 5, 7]] == 0.0) and np.all(M[:, 8] == 1.0)
    a = re.search(r"const float tabAmplitude\[size\*size\] = \{(.*?)\n\};", src, re.S)
    amp = np.array([float(x.strip().rstrip("f")) for x in a.group(1).split(",") if x.strip()],
                   dtype=np.float64)
    assert amp.shape == (4096,), amp.shape
    rec = np.stack([M[:, 0], M[:, 2], M[:, 4], M[:, 6], amp], axis=1).astype(np.float32)
    return rec


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name in ("ggx", "beckmann"):
        rec = extract(name)
        path = os.path.join(OUT, f"ltc_{name}.f32")
        rec.tofile(path)
        print(name, rec.shape, "m0", rec[:, 0].min(), rec[:, 0].max(), "m6", rec[:, 3].min(),
              rec[:, 3].max(), "amp", rec[:, 4].min(), rec[:, 4].max(), "->", path)
