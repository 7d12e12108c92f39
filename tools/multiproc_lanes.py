"""Is batch-1 throughput bounded by the FOUR hardware queues of a process or by the GPU?  N processes on ONE GPU, each with
its own 4 queues and `lanes` captured forwards, start a timed loop together (a multiprocessing barrier) and submit for a fixed
number of steps; prints each process's and the summed rate.   python tools/multiproc_lanes.py [N processes ...]"""
import importlib, multiprocessing as mp, sys, time


def worker(rank, n, lanes, steps, barrier, out):
    import torch
    sys.path.insert(0, ".")
    pkg = lambda m: importlib.import_module("efficientlo-net_amd." + m)
    model, synth = pkg("model"), pkg("synth")
    dev = torch.device("cuda:0")
    net = model.PWCLONet(dev, seed=0)
    f1, f2 = synth.frame_pair(1, 64, 1800, seed=1 + rank)
    pair = torch.cat([torch.from_numpy(f1), torch.from_numpy(f2)], 0).to(dev)
    net.capture(1, 64, 1800, lanes=lanes, pose_ring=64)
    for i in range(4 * lanes):
        net.submit(i % lanes, pair)
    torch.cuda.synchronize()
    barrier.wait()
    t0 = time.perf_counter()
    for i in range(steps):
        net.submit(i % lanes, pair)
    torch.cuda.synchronize()
    out.put((rank, steps / (time.perf_counter() - t0)))


if __name__ == "__main__":
    mp.set_start_method("spawn")
    counts = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
    for n in counts:
        for lanes in (4, 8):
            barrier, out = mp.Barrier(n), mp.Queue()
            procs = [mp.Process(target=worker, args=(r, n, lanes, 6000, barrier, out)) for r in range(n)]
            for p in procs: p.start()
            rates = sorted(out.get() for _ in procs)
            for p in procs: p.join()
            print("%d process(es) x %d lanes: %s -> %.0f pairs/s in all" % (n, lanes, " ".join("%.0f" % r for _, r in rates), sum(r for _, r in rates)), flush=True)
