"""oracle/cpu_bench.py -- the timed CPU baseline of bench.py (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

The reference has no CPU implementation (fused_conv.cpp:176 registers a GPU kernel only, the Python needs
TensorFlow): "the reference CPU path" is the reference ALGORITHM on host cores, i.e. this package's restatement
(oracle/ops_np.get_model_from_projection: numpy fp32 + the C grouping oracle), kind "port".

Three legs on the same bounded sample:
  * one core: a single process, BLAS limited to one thread;
  * all cores: one single-threaded worker PROCESS per host core, each running whole frame pairs (frame pairs are
    independent, so this is the same data-parallel split the GPUs use) -- spawned workers that import numpy only;
  * one process, all threads: SURVEY 8(d)'s literal recipe -- ONE process, the C grouping oracle with a thread per
    core over the centres (ELO_ORACLE_THREADS) and the dense layers on the BLAS's thread pool -- which is the weaker of
    the two all-core readings (a pair is ~0.2 s of small operators: threads per operator scale worse than pairs per
    core), reported beside it for completeness.
Inputs (weights, one frame pair) travel through an .npz file so that the workers never import torch.
"""
import os
import sys
import time


def _limit_threads():
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[v] = "1"


def _shuffle_table():
    import numpy as np
    table = {}

    def shuffle(scope, tag, KT):
        key = (scope, tag, KT)
        if key not in table:
            table[key] = np.random.default_rng(len(table)).permutation(KT).astype(np.int32)
        return table[key]
    return shuffle


def worker(npz_path, pairs, start_at, threads=1):
    """Run `pairs` forwards; returns (first start, last end) wall-clock stamps.  Spins until `start_at` so that all
    workers of the all-cores leg run their pairs at the same time.  threads > 1: the threaded leg (set before numpy loads)."""
    if threads > 1:
        for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            os.environ[v] = str(min(threads, 64))              # (BLAS thread pools above 64 only add contention on 42-row GEMMs)
        os.environ["ELO_ORACLE_THREADS"] = str(threads)
    else:
        _limit_threads()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import numpy as np
    from oracle import ops_np as O
    blob = np.load(npz_path)
    params = {k[2:]: blob[k] for k in blob.files if k.startswith("p:")}
    f1, f2 = blob["f1"], blob["f2"]
    shuffle = _shuffle_table()
    O.get_model_from_projection(params, shuffle, f1, f2)          # untimed: page the libraries in, build the tables
    while time.time() < start_at:
        pass
    t0 = time.time()
    for _ in range(pairs):
        O.get_model_from_projection(params, shuffle, f1, f2)
    return t0, time.time()


def run(params, f1, f2, pairs_one_core, pairs_per_worker, workers=None):
    """-> dict(one_core=pairs/s, all_cores=pairs/s, cores=workers, seconds=(t1, tN))."""
    import multiprocessing as mp
    import tempfile

    import numpy as np
    workers = workers or os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "cpu_bench.npz")
        np.savez(path, f1=f1, f2=f2, **{"p:" + k: v for k, v in params.items()})
        ctx = mp.get_context("spawn")
        with ctx.Pool(1) as pool:                                    # the one-core leg, in a clean single-threaded process
            a, b = pool.apply(worker, (path, pairs_one_core, 0.0))
        t_one = b - a
        with ctx.Pool(workers) as pool:
            start_at = time.time() + 4.0 + 0.1 * workers               # interpreter start + the untimed warm-up forward (a late
                                                                       # worker starts late: the span below still covers it)
            spans = pool.starmap(worker, [(path, pairs_per_worker, start_at)] * workers)
        t_all = max(e for _, e in spans) - min(s for s, _ in spans)
        with ctx.Pool(1) as pool:                                    # SURVEY 8(d)'s recipe: one process, threads inside the operators
            pairs_threaded = max(4, pairs_one_core // 4)
            a, b = pool.apply(worker, (path, pairs_threaded, 0.0, workers))
        t_thr = b - a
    return {"one_core": pairs_one_core / t_one, "all_cores": workers * pairs_per_worker / t_all, "cores": workers,
            "one_process_all_threads": pairs_threaded / t_thr, "seconds": (t_one, t_all, t_thr), "pairs_threaded": pairs_threaded}
