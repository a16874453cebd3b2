"""Worker of bench.py's all-core CPU baseline (test / measurement infrastructure, like the rest of
``oracle/``: never imported by the product).  Each process runs the numpy/scipy restatement of the
path (``hotpath.run_path``: atmosphere/atmosphere.py:293-380, sim/atmosphere.py:39-84) on its own
detector rows of the same synthetic configuration and reports how long that took."""

from __future__ import annotations

import os
import time


def worker(config, n_total, index, n_rows, screens_path, barrier, queue, block=64):
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    try:
        import numpy as np

        from maria_amd import synthetic
        from oracle import hotpath

        problem = synthetic.config_problem(config, n_det=n_total)
        screens = np.load(screens_path, mmap_mode="r")
        rows = (index * n_rows + np.arange(n_rows)) % len(problem["offsets"])
        sub = dict(problem)
        sub["layers"] = [dict(l, values=np.asarray(s)) for l, s in zip(problem["layers"], screens)]
        checksum = 0.0
        barrier.wait(timeout=300)
        t0 = time.perf_counter()
        for a in range(0, n_rows, block):  # small blocks: the float64 intermediates of 64 processes must fit the host
            blk = dict(sub)
            for key in ("offsets", "band_index", "m00", "gain"):
                blk[key] = None if problem.get(key) is None else problem[key][rows[a : a + block]]
            checksum += float(hotpath.run_path(blk)[:, ::4096].sum())
        queue.put({"index": index, "seconds": time.perf_counter() - t0, "checksum": checksum})
    except Exception as exc:  # report instead of hanging the parent's queue.get
        try:
            barrier.abort()
        except Exception:
            pass
        queue.put({"index": index, "error": f"{type(exc).__name__}: {exc}"[:300]})
