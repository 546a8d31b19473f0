#!/usr/bin/env python3
"""Accuracy of fast_rcp / fast_rsqrt / sqrt_from_rsqrt of the library named by SFHIP_LIBRARY (sfhip_selftest_fastmath):
max relative error in units of 2^-52 over 30 decades.  With a -DSFHIP_RSQ_RAW build this is the hardware seeds' accuracy."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sailfish_amd.kernel.library import Library  # noqa: E402

lib = Library(mode="hip", debug=False)
rng = np.random.default_rng(0)
x = np.concatenate([10.0 ** rng.uniform(-15, 15, 400000), rng.uniform(0.5, 2.0, 400000)])
n = x.size
out = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
lib.sfhip_selftest_fastmath[n](torch.from_numpy(x).cuda(), out)
got = out.cpu().numpy().reshape(3, n)
xl = x.astype(np.longdouble)
for name, g, exact in zip(("rcp", "rsqrt", "sqrt"), got, (1 / xl, 1 / np.sqrt(xl), np.sqrt(xl))):
    rel = np.abs((g.astype(np.longdouble) - exact) / exact).max()
    print(f"{os.environ.get('SFHIP_LIBRARY', 'default')} {name}: max relative error {float(rel):.3e} = 2^{np.log2(float(rel)):.1f}")
