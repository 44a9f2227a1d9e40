#!/usr/bin/env python3
"""Benchmark of the Consenrich estimator hot path on MI355X (driver contract: one JSON line on rank 0).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "genomic bins/sec (forward+backward pass), hg38 200bp x 32 samples"):
22 synthetic chains with the hg38 autosome bin counts at 200 bp (14 375 018 bins), m = 32 samples, levelTrend model,
SURVEY 8(d) parameters.  One STEP = one full pass of the hot path over every chain the rank owns, inputs already
resident in HBM:  per-bin sufficient statistics of (data, munc)  ->  forward filter (store, NLL)  ->  RTS smoother
->  lag-one covariances  ->  D, xf, Pf, pNoise, xs, Ps, lagCov in the reference layouts  ->  residuals (n, m).
With N > 1 the chains are LPT-sharded over the ranks (strong scaling: the genome is fixed); there is no data-path
collective -- the one RCCL call is the final track gather, done once after the timed region and reported separately.

No PyTorch: the launcher (`python -m torch.distributed.run`) only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT in
the environment; the barrier and the max-over-ranks of the timed region are RCCL all-reduces on the library's stream
(consenrich_amd.sharding.RcclComm -> csr_comm_* in libconsenrich_amd.so, librccl dlopen'ed there), the rendezvous of the
128-byte RCCL id is a file on the node.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
TRAFFIC_FILE = os.path.join("profiles", "r02_pmc_traffic.json")     # PMC pass of THIS workload (scripts/pmc_traffic.py)


def b_alg(m: int) -> int:
    """algorithmic bytes per bin, forward + backward, data read once."""
    return 12 * m + 100


# algorithmic bytes per bin of each kernel (DESIGN.md "Kernels"): only traffic the reference layouts require
def kernel_alg_bytes(name: str, m: int, d: int) -> float:
    return {
        "stats": 8.0 * m,                       # data + munc read once
        "residuals": 4.0 * m,                   # (n, m) residual write
        "fwd_chain": 16.0 + 16.0 + 16.0 + 8.0,  # lambda/kappa/qscale/blockMap in, Pf + pNoise + xf out (fused chain)
        "fwd_cov_chain": 16.0 + 16.0 + 16.0,    # lambda/kappa/qscale/blockMap in, Pf + pNoise out
        "fwd_state_chain": 8.0,                 # xf out
        "fwd_dstat": 4.0,                       # D out
        "bwd_chain": 8.0 + 16.0 + 16.0,         # xs + Ps + lagCov out
        "export_natural": 0.0,                  # layout conversion: pure overhead
    }.get(name, 0.0)


class FileComm:
    """Control-plane fallback over files on the node (barrier and max only): used to agree on whether RCCL is usable on
    every rank, and in its place if it is not -- so that a broken RCCL installation costs the gather, not the measurement."""

    def __init__(self, rank: int, world: int):
        self.rank, self.world, self.k = rank, world, 0
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
        self.dir = os.path.join(base, f"consenrich_amd_fc_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
        os.makedirs(self.dir, exist_ok=True)

    def allreduce_max(self, value: float, timeout_s: float = 600.0) -> float:
        self.k += 1
        mine = os.path.join(self.dir, f"{self.k}_{self.rank}")
        with open(mine + ".tmp", "w") as fh:
            fh.write(repr(float(value)))
        os.replace(mine + ".tmp", mine)
        vals, deadline = [], time.monotonic() + timeout_s
        for r in range(self.world):
            path = os.path.join(self.dir, f"{self.k}_{r}")
            while True:
                try:
                    with open(path) as fh:
                        vals.append(float(fh.read()))
                    break
                except (FileNotFoundError, ValueError):
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rank {self.rank}: rank {r} never reached barrier {self.k}")
                    time.sleep(0.0005)
        return max(vals)

    def barrier(self):
        self.allreduce_max(0.0)

    def close(self):
        self.barrier()
        if self.rank == 0:
            time.sleep(0.2)
            shutil.rmtree(self.dir, ignore_errors=True)


def cpu_baseline(m: int, max_seconds: float = 15.0):
    """Oracle (C port of the reference loop, 1 thread) on a bounded sample of the same workload: a chr1-sized chain.
    (scripts/cpu_port_vs_reference.py, build container only: the port runs within 3 % of the compiled reference.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases
    from consenrich_amd.sharding import hg38_chain_lengths
    from oracle import oracle as orc

    orc.lib()
    n = hg38_chain_lengths(200)[0]    # chr1-sized chain: 1 244 783 bins
    data, munc = cases.synth(n, m, 21)
    F = np.asarray(cases.F_TREND, np.float32)
    Q0 = np.diag([1e-3, 1e-4]).astype(np.float32)
    bm = (np.arange(n) // 500).astype(np.int32)
    xf, Pf, pn = np.empty((n, 2), np.float32), np.empty((n, 2, 2), np.float32), np.empty((n, 2, 2), np.float32)
    D = np.empty(n, np.float32)
    xs, Ps = np.empty((n, 2), np.float32), np.empty((n, 2, 2), np.float32)
    lag, res = np.empty((n - 1, 2, 2), np.float32), np.empty((n, m), np.float32)

    def one():
        orc.cforwardPass(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0, intervalToBlockMap=bm,
                         blockCount=int(bm.max()) + 1, stateInit=0.0, stateCovarInit=1000.0, stateForward=xf,
                         stateCovarForward=Pf, pNoiseForward=pn, vectorD=D, returnNLL=True)
        orc.cbackwardPass(matrixData=data, matrixF=F, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn,
                          stateSmoothed=xs, stateCovarSmoothed=Ps, lagCovSmoothed=lag, postFitResiduals=res)

    one()  # warm-up (first-touch page faults)
    best, spent, reps = float("inf"), 0.0, 0
    while reps < 400 and spent < max_seconds:
        t = time.perf_counter()
        one()
        dt = time.perf_counter() - t
        best = min(best, dt)
        spent += dt
        reps += 1
    return {
        "value": n / best, "unit": "genomic bins/s", "cores": 1, "kind": "port",
        "sample": f"oracle C port (forward store+NLL, backward+residuals), 1 thread, chr1-sized chain "
                  f"({n} bins x {m} samples), best of {reps} passes after warm-up ({spent:.1f} s of CPU work)",
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--bin-bp", type=int, default=200)
    ap.add_argument("--q0", default="1e-3,1e-4", help="diagonal of the base process noise Q0 (long-memory regime: 1e-5,1e-6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact-mode and ECM measurements")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0 (no RCCL: file barrier)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        print("bench.py: --gpus N > 1 must be launched through torch.distributed.run (one process per GPU)", file=sys.stderr)
        return 2

    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams
    from consenrich_amd.sharding import RcclComm, hg38_chain_lengths, lpt_assign

    m = args.samples
    q00, q11 = (float(v) for v in args.q0.split(","))
    lengths = hg38_chain_lengths(args.bin_bp)
    total_bins = int(sum(lengths))
    mine = lpt_assign(lengths, world)[rank]
    my_lens = [lengths[i] for i in mine]

    model = ModelParams(state_dim=2, Q0=((q00, 0.0), (0.0, q11)))
    batch = DeviceBatch(local_rank)
    comm, comm_kind, comm_note, fc, rccl_hung = None, "none", None, None, False
    if world > 1:
        fc = FileComm(rank, world)
        probe_ok = 0.0
        if args.same_device:
            probe_ok, comm_note = 1.0, "same-device rehearsal: RCCL refuses two ranks on one GPU"
        else:
            import ctypes

            if L.lib().csr_comm_unique_id(ctypes.create_string_buffer(128)) != 0:      # RCCL loads and sees this device?
                probe_ok, comm_note = 1.0, f"rank {rank}: {L.last_error()}"
        rccl_hung = False
        if fc.allreduce_max(probe_ok) == 0.0:
            # communicator creation is a collective: run it under a watchdog so that a bootstrap that never completes on this
            # node costs the gather, not the measurement (the ranks then agree, through the files, to use the file barrier)
            import threading

            box = {}

            def make():
                try:
                    box["comm"] = RcclComm(batch, world, rank)
                except Exception as exc:        # noqa: BLE001
                    box["err"] = repr(exc)

            th = threading.Thread(target=make, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get("CONSENRICH_AMD_RCCL_TIMEOUT", "240")))
            mine_ok = "comm" in box
            if fc.allreduce_max(0.0 if mine_ok else 1.0) == 0.0:
                comm, comm_kind = box["comm"], "rccl"
            else:
                comm, comm_kind = fc, "file"
                rccl_hung = th.is_alive()
                comm_note = f"rank {rank}: RCCL communicator not created ({box.get('err', 'timeout')})" if not mine_ok \
                    else "RCCL communicator not created on another rank"
        else:
            comm, comm_kind = fc, "file"
            comm_note = comm_note or "RCCL unusable on another rank"
    batch.configure(model, m, my_lens)
    batch.synthesize(seed=1234 + rank)
    flags = L.RETURN_NLL
    what = L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID

    def fence(b=batch):
        # device-wide synchronize of this rank, then the barrier over all ranks (an RCCL all-reduce on the library's
        # stream + stream synchronisation), then nothing is in flight anywhere
        b.synchronize()
        if comm is not None:
            comm.barrier()

    def max_over_ranks(v: float) -> float:
        return comm.allreduce_max(v) if comm is not None else v

    def timed(b, fn, warmup, steps):
        for _ in range(warmup):
            fn()
        fence(b)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        fence(b)
        return max_over_ranks(time.perf_counter() - t0)

    # statistics + forward (store, NLL) + backward + every track in the reference layout + residuals + per-chain phiHat /
    # NLL read-back (the step's one host synchronisation), as one C-ABI call
    elapsed = timed(batch, lambda: batch.step(flags, what), args.warmup, args.steps)
    ms_per_step = 1000.0 * elapsed / max(args.steps, 1)
    value = total_bins * args.steps / elapsed

    # per-kernel durations: HIP events on the library's stream, separate (untimed) pass of the same steps
    batch.profile(True)
    for _ in range(args.steps):
        batch.step(flags, what)
    times = batch.kernel_times()
    batch.profile(False)
    rs = batch.run_stats()
    my_bins = int(sum(my_lens))
    per_kernel = {k: {"launches": v[0], "avg_ms": v[1] / max(v[0], 1), "ms_per_step": v[1] / max(args.steps, 1)}
                  for k, v in times.items()}
    # dominant kernel = the longest one ON THE CRITICAL PATH: the NIS/NLL epilogue runs on the side stream underneath the
    # smoother / residual kernels (its event-measured duration is stretched by that overlap), so it never is
    critical = {k: v for k, v in per_kernel.items() if k != "fwd_dstat"} or per_kernel
    dom = max(critical, key=lambda k: critical[k]["ms_per_step"]) if critical else None
    roofline = None
    if dom is not None:
        alg_bytes = kernel_alg_bytes(dom, m, 2) * my_bins
        avg_s = per_kernel[dom]["avg_ms"] * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, TRAFFIC_FILE)
        # the committed PMC pass was taken on the default workload at N = 1; it does not describe other shapes / shards
        if os.path.exists(pmc_path) and world == 1 and m == 32 and args.bin_bp == 200 and args.q0 == "1e-3,1e-4":
            try:
                with open(pmc_path) as fh:
                    traffic = json.load(fh).get(dom, {}).get("hbm_bytes_per_launch")
                traffic_source = f"{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes of this " \
                                 "command; not measured in this run)"
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                    "alg_bytes_per_bin": kernel_alg_bytes(dom, m, 2), "bins_per_launch": my_bins,
                    "avg_launch_ms": per_kernel[dom]["avg_ms"]}

    extras = {}
    if not args.no_extras:
        # (1) the ECM loop -- what real runs execute: 5 x [forward, smoother + kappa E-step] + 1 NLL forward
        ecm_iters, inner = 3, 5
        batch.stats()

        def ecm_once():
            batch.ecm(max_iters=ecm_iters, inner_iters=inner, rtol=0.0, use_lambda=False, use_kappa=True)

        e = timed(batch, ecm_once, 1, 2) / 2.0
        rs_ecm = batch.run_stats()
        extras["ecm"] = {"ms_per_iter": 1000.0 * e / ecm_iters, "iters": ecm_iters, "inner_sweeps": inner,
                         "bin_sweeps_per_s": total_bins * ecm_iters * inner / e,
                         "pipeline_redos": rs_ecm["pipeline_redos"] - rs["pipeline_redos"],
                         "warm_started_sweeps": {"active": rs_ecm["block_len"] <= 32, "windows_bins": [rs_ecm["ws_warm_f"], rs_ecm["ws_warm_b"]],
                                                 "blocks_repaired_in_kernel": rs_ecm["local_repairs"],
                                                 "note": "batches with 32-bin blocks (< 2 M bins per rank: 8-GPU shards) start a "
                                                         "sweep's windows from the previous sweep's carries"},
                         "note": "per ECM iteration over all chains: 5 x (forward + smoother + kappa E-step) + 1 NLL "
                                 "forward; bin_sweeps = forward+backward+E-step sweeps"}
        # (2) the same step in the bit-exact validation mode (k = 0: results == the sequential recursion; the mode the
        # reference-shaped drop-in callables default to)
        ex = DeviceBatch(local_rank, x_tol_ulps=0)
        ex.configure(model, m, my_lens)
        ex.synthesize(seed=1234 + rank)
        ex_steps = max(1, min(args.steps, 3))
        ee = timed(ex, lambda: ex.step(flags, what), 1, ex_steps)
        rx = ex.run_stats()
        extras["exact_mode"] = {"x_tol_ulps": 0, "ms_per_step": 1000.0 * ee / ex_steps, "value": total_bins * ex_steps / ee,
                                "unit": "genomic bins/s", "steps": ex_steps, "block_len": rx["block_len"],
                                "warm_bins": [rx["warm_p"], rx["warm_x"], rx["warm_b"]],
                                "state_chain": {"superblock_bins": int(os.environ.get("CONSENRICH_AMD_SB_BINS", "8192")),
                                                "window_bins": int(os.environ.get("CONSENRICH_AMD_SB_WARM", "16384")),
                                                "note": "bitwise speculation of the state chain on a re-blocked view; "
                                                        "reruns[1] counts superblocks re-run over 1 + steps steps"},
                                "reruns": [rx["reruns_p"], rx["reruns_x"], rx["reruns_b"]],
                                "fix_launches": rx["fix_launches"]}
        ex.close()
        # (3) the same step with the delete-block calibration folds of every chromosome as extra chains of the batch
        # (uncertainty.py:1370-1419: folds = 2 independent refits per chromosome, constants.py:437; DeviceBatch.make_fold):
        # what fills an under-occupied shard -- the latency-bound chain kernels take about as long for 3x the chains
        folds = 2
        fb = DeviceBatch(local_rank)
        fb.configure(model, m, [n for n in my_lens for _ in range(folds + 1)])
        fb.synthesize(seed=4321 + rank)
        ef = timed(fb, lambda: fb.step(flags, what), 1, max(1, min(args.steps, 5)))
        fsteps = max(1, min(args.steps, 5))
        extras["with_calibration_folds"] = {"folds": folds, "ms_per_step": 1000.0 * ef / fsteps,
                                            "value": total_bins * (folds + 1) * fsteps / ef, "unit": "genomic bins/s",
                                            "note": "every chromosome three times in the batch (fit + 2 fold refits, "
                                                    "synthetic data of the same shape); bins of all refits counted"}
        fb.close()

    gather_ms, gather_note = None, comm_note
    if comm_kind == "rccl" and not args.no_gather:
        # final track gather (state + its variance): once per job, packed on the device from the exported arrays and
        # all-gathered over RCCL / xGMI; not part of `value`.  A failure here must not cost the measurement its JSON line.
        try:
            batch.step(flags, what)
            fence()
            tg = time.perf_counter()
            gathered = comm.gather_batch_tracks(lengths, to_host=True)
            fence()
            gather_ms = 1000.0 * max_over_ranks(time.perf_counter() - tg)
            if rank == 0 and not all(g is not None and g.shape == (lengths[i], 2) and np.all(np.isfinite(g))
                                     for i, g in enumerate(gathered)):
                gather_note = "gathered tracks have unexpected shapes / values"
        except Exception as exc:        # noqa: BLE001
            gather_note = f"gather failed: {exc!r}"

    if rank == 0:
        out = {
            "metric": "genomic bins/sec (forward+backward pass), hg38 200bp x 32 samples",
            "value": value, "unit": "genomic bins/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64 arithmetic on f32 storage", "data": "synthetic",
            "config": {
                "workload": f"hg38 autosomes, 22 chains / {total_bins} bins @{args.bin_bp}bp x {m} samples; "
                            "stats + forward(store,NLL) + RTS backward + lagCov + reference-layout tracks + residuals, "
                            "levelTrend",
                "chains_per_rank": "LPT over contigs", "block_len": rs["block_len"], "q0_diag": [q00, q11],
                "warm_bins": [rs["warm_p"], rs["warm_x"], rs["warm_b"]], "x_tol_ulps": rs["x_tol_ulps"],
                "comm": comm_kind,
            },
            "roofline": roofline,
            "path_roofline": {"alg_bytes_per_bin": b_alg(m), "achieved": value * b_alg(m) / 1e9,
                              "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                              "frac": value * b_alg(m) / 1e9 / (HBM_PEAK_GBS * world)},
            "kernels_rank0": per_kernel,
            "speculation": {"blocks": rs["blocks"], "reruns_cov": rs["reruns_p"], "reruns_state": rs["reruns_x"],
                            "reruns_bwd": rs["reruns_b"], "fix_launches": rs["fix_launches"],
                            "pipeline_redos": rs["pipeline_redos"]},
            "gather_ms": gather_ms,
        }
        out.update(extras)
        if gather_note:
            out["gather_note"] = gather_note
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(m)
        print(json.dumps(out))
    if comm is not None:
        comm.barrier()
        if comm is not fc:
            comm.close()
    if fc is not None:
        fc.close()
    if rccl_hung:               # a thread is still stuck inside ncclCommInitRank: do not wait for it at interpreter exit
        sys.stdout.flush()
        os._exit(0)
    batch.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
