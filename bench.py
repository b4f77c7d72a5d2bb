#!/usr/bin/env python3
"""bench.py -- juliet pileup + Fisher hot path on MI355X: aligned CCS reads/s and % of the HBM roofline.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pol_1m] [--reads R]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path over one batch of synthetic reads already resident in HBM:
window bounds -> zero the count matrices -> pileup (nucleotide + codon counts) -> [N > 1: RCCL all-reduce of the
integer count matrices] -> CountNumberOfTests + per-(position, codon) Fisher cells.  By default the window bounds
stay in HBM between the bounds kernel and the pileup (msq_window_bounds_device / msq_pileup_device_window: the
count matrices are sized from the set-up window, a wider window would raise a deferred error, and the bounds are
compared with the set-up window after the timed loop); --host-bounds brings them back to the host every step, as
MSAByRow's constructor does.  N > 1 is weak scaling:
every rank holds its own shard of `reads` reads of one global sample (global read index = rank * reads + i),
there is no data-path collective besides the count all-reduce the north star prescribes.

Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel (the read-streaming pileup kernel) timed with
HIP events on its own stream inside the timed region; `cpu_baseline` is the oracle (CPU restatement of the
reference path) on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pol_1m", help="pol_1m (default: 1M x 2843, north-star target), pol_100k, "
                    "hiv_mix_200k, amp20k_10m")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the workload's)")
    ap.add_argument("--qualqv", type=int, default=-1, help="QUALQV threshold (env var of the reference); -1 = unset")
    ap.add_argument("--rich", action="store_true", help="reads carry DelQV / SubQV / InsQV as well (4.5 B/base) and the "
                    "reference's default SUBQV=42 applies (src/QvThresholds.cpp:45-56)")
    ap.add_argument("--no-bake", action="store_true", help="with --qualqv: keep the QV plane and apply the threshold "
                    "inside every pileup pass instead of once at set-up (msq_reads_bake_qv)")
    ap.add_argument("--mode", default="auto", choices=["auto", "simple", "stream"])
    ap.add_argument("--host-bounds", action="store_true", help="bring the window bounds back to the host every step "
                    "(msq_window_bounds + msq_pileup) instead of leaving them in HBM (msq_window_bounds_device + "
                    "msq_pileup_device_window, the default)")
    ap.add_argument("--no-partition", action="store_true", help="windows wider than 4096 columns: run every slab launch "
                    "over the whole read set instead of over the column-slab partition built at set-up")
    ap.add_argument("--slab-columns", type=int, default=0, help="widest column slab of the partition (multiple of 32, "
                    "<= 4096); narrower than the window: partition even a window that fits one slab")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: initialise the RCCL process group and issue "
                    "the step's collectives even with one rank")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="reads of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-worker", default="", help=argparse.SUPPRESS)  # name:first:n -> prints the oracle's seconds
    ap.add_argument("--h2d-steps", type=int, default=0, help="also time K steps that first copy the packed reads from "
                    "pinned host memory (PCIe-inclusive rate, reported beside `value`, never as it)")
    return ap.parse_args()


def load_traffic(kernel, workload, reads, qualqv, baked=False):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic.json; rocprofv3 cannot run inside the
    timed process), or None when no pass was recorded for this exact kernel / workload / size / QV setting."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            for e in json.load(f)["entries"]:
                if (e["kernel"] == kernel and e["workload"] == workload and e["reads"] == reads
                        and e.get("qualqv") == (None if qualqv < 0 else qualqv) and bool(e.get("baked")) == baked):
                    return e["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def _cpu_worker(args):
    """One CPU-baseline process: the oracle's juliet path on reads [first, first + n); returns its compute seconds."""
    name, first, n = args
    from minorseq_amd import synth
    from oracle import orc
    wl = synth.workload(name)
    reads = wl.host_reads(first, n)  # untimed: the reference, too, is timed from AminoAcidCaller's constructor on
    gb, ge = wl.gene_arrays()
    t0 = time.perf_counter()
    m = orc.Msa(reads)  # MSAByRow + MSAByColumn
    m.call(gb, ge, wl.reference if wl.use_reference else "")  # CountNumberOfTests + CallVariants' cell loop
    dt = time.perf_counter() - t0
    m.close()
    return dt


def cpu_baseline(name, sample):
    """The oracle (CPU restatement of the reference path; the reference itself is single-threaded) on a bounded
    sample: one core on `sample` reads, then every host core this job may use as independent processes on
    sample / 4 reads each (processes, not threads: src/Fisher.cpp:117,131-134 keep statics).  The workers are
    fresh interpreters (`bench.py --cpu-worker`), started before this process loads the HIP library or touches
    the GPU."""
    import subprocess

    def run(jobs):
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "%s:%d:%d" % j],
                               stdout=subprocess.PIPE, text=True) for j in jobs]
        outs = [p.communicate()[0] for p in ps]
        if any(p.returncode for p in ps):
            raise SystemExit("cpu baseline worker failed")
        return [float(o.strip().splitlines()[-1]) for o in outs]

    dt1 = run([(name, 0, sample)])[0]
    procs = max(1, min(16, os.cpu_count() or 1))  # 16 = the CPU share of a one-GPU box
    per = max(1000, sample // 4)
    dts = run([(name, k * per, per) for k in range(procs)])
    return {"value": sample / dt1, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of the same synthetic workload, %.1f s on one host core (oracle/liborc.so)" % (sample, dt1),
            "all_cores": {"value": procs * per / max(dts), "unit": "reads/s", "cores": procs,
                          "sample": "%d processes x %d reads, slowest %.1f s" % (procs, per, max(dts))}}


def main():
    a = parse()
    if a.cpu_worker:
        name, first, n = a.cpu_worker.split(":")
        print(_cpu_worker((name, int(first), int(n))))
        return
    cpu = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.cpu_sample > 0:
        from minorseq_amd import synth as _synth
        cpu = cpu_baseline(a.workload, min(a.cpu_sample, a.reads or _synth.workload(a.workload).n_reads))
    import torch
    import torch.distributed as dist
    from minorseq_amd import abi, device, sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1 or a.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        sharding.FORCE_COLLECTIVES = a.force_dist

    wl = synth.workload(a.workload, rich_qv=True) if a.rich else synth.workload(a.workload)
    reads = a.reads or wl.n_reads
    ctx = device.Ctx(local)
    L = abi.lib()
    thr = abi.QvThresholds.make(qual=a.qualqv, sub=42 if a.rich else -1)
    mode = {"auto": abi.MSQ_PILEUP_AUTO, "simple": abi.MSQ_PILEUP_SIMPLE, "stream": abi.MSQ_PILEUP_STREAM}[a.mode]

    # ---- setup (untimed): reads into HBM, plan, log-factorial table
    dr = device.synth_device(ctx, wl, rank * reads, reads)
    n_bases = dr.n_groups * 32
    # a run's QV thresholds are fixed before the reads are loaded: they are applied once, at pack time (untimed like
    # the packing itself), and the passes below stream the baked set -- as the front does (MSAByRow::Finish)
    baked = (a.qualqv >= 0 or a.rich) and not a.no_bake and a.h2d_steps == 0
    if baked:
        src = dr
        dr = device.bake_qv(ctx, src, thr)
        ctx.sync()
        del src
    b0, e0 = sharding.allreduce_bounds(*device.window_bounds(ctx, dr), "cuda")
    W = e0 - b0
    gb, ge = wl.gene_arrays()
    plan = device.Plan(ctx, gb, ge, wl.reference if wl.use_reference else "", b0 + 1, W)
    ctx.lfact_reserve(2 * reads * world + 2)
    counts = torch.zeros(W * 70, dtype=torch.int32, device="cuda")  # [W][6] nucleotide then [W][64] codon, one buffer
    nuc, codon = counts[:W * 6], counts[W * 6:]
    em = abi.ErrorModel()
    abi.check(L.msq_error_model_default(C.byref(em)))

    bounds = torch.zeros(2, dtype=torch.int32, device="cuda")  # {-begin0, end0}, device-resident

    def step_host_bounds():  # the window comes back to the host every step, as MSAByRow's constructor needs it
        sharding.allreduce_bounds(*device.window_bounds(ctx, dr), "cuda")  # MSAByRow pass 1 (UpdateBoundaries)
        counts.zero_()
        abi.check(L.msq_pileup(ctx.h, dr.view, thr, b0, W, nuc.data_ptr(), codon.data_ptr(), mode))
        sharding.allreduce_counts(counts)  # N > 1: RCCL sum of the integer count matrices over xGMI
        abi.check(L.msq_call_cells(ctx.h, plan.h, codon.data_ptr(), C.byref(em), 100.0, plan.positions.data_ptr(),
                                   plan.p.data_ptr(), plan.expected.data_ptr(), plan.n_tests.data_ptr()))

    def step_device_bounds():  # the same work with the window left in HBM: nothing in the step waits for the host
        abi.check(L.msq_window_bounds_device(ctx.h, dr.view, bounds.data_ptr()))
        sharding.allreduce_bounds_device(bounds)  # N > 1: one 2-int MAX
        counts.zero_()
        abi.check(L.msq_pileup_device_window(ctx.h, dr.view, thr, bounds.data_ptr(), b0, W, nuc.data_ptr(), codon.data_ptr()))
        sharding.allreduce_counts(counts)
        abi.check(L.msq_call_cells(ctx.h, plan.h, codon.data_ptr(), C.byref(em), 100.0, plan.positions.data_ptr(),
                                   plan.p.data_ptr(), plan.expected.data_ptr(), plan.n_tests.data_ptr()))

    # windows wider than one slab of the streaming kernel: the reads are laid out per column slab at set-up (pack
    # time, untimed like the packing itself), so that every slab launch streams only its share of the reads
    part = None
    if (W > 4096 or 0 < a.slab_columns < W) and not a.no_partition and a.mode != "simple":
        part = device.Partition(ctx, dr, b0, W, a.slab_columns)

    def step_partition():
        abi.check(L.msq_window_bounds_device(ctx.h, dr.view, bounds.data_ptr()))
        sharding.allreduce_bounds_device(bounds)
        counts.zero_()
        abi.check(L.msq_pileup_partition(ctx.h, part._h, thr, nuc.data_ptr(), codon.data_ptr()))
        sharding.allreduce_counts(counts)
        abi.check(L.msq_call_cells(ctx.h, plan.h, codon.data_ptr(), C.byref(em), 100.0, plan.positions.data_ptr(),
                                   plan.p.data_ptr(), plan.expected.data_ptr(), plan.n_tests.data_ptr()))

    step = step_partition if part else step_host_bounds if (a.host_bounds or a.mode == "simple") else step_device_bounds
    launches_per_step = part.n_slabs if part else 1  # timing slots: one per pileup call that reaches the kernel

    def barrier():
        if world > 1 or a.force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    abi.check(L.msq_pileup_check(ctx.h))
    abi.check(L.msq_call_check(ctx.h))
    abi.check(L.msq_timing_enable(ctx.h, a.steps * launches_per_step))
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    abi.check(L.msq_pileup_check(ctx.h))
    abi.check(L.msq_call_check(ctx.h))
    if step is not step_host_bounds and bounds.tolist() != [-b0, e0]:
        raise SystemExit("device-resident window bounds %s differ from the set-up window [%d, %d)" % (bounds.tolist(), b0, e0))
    n_slots = a.steps * launches_per_step
    ms = (C.c_float * n_slots)()
    n_ms = C.c_int32(0)
    abi.check(L.msq_timing_read(ctx.h, ms, n_slots, C.byref(n_ms)))
    abi.check(L.msq_timing_enable(ctx.h, 0))
    kernel_ms = sum(ms[:n_ms.value]) / max(1, a.steps)  # per step: the slab launches of a partition added up

    pcie = None
    if a.h2d_steps > 0 and world == 1:  # the boundary handing over host buffers: packed reads cross PCIe every step
        host = {f: t.cpu().pin_memory() for f, t in dr.t.items() if t is not None}
        barrier()
        t1 = time.perf_counter()
        for _ in range(a.h2d_steps):
            for f, h in host.items():
                dr.t[f].copy_(h, non_blocking=True)
            step()
        barrier()
        dt1 = (time.perf_counter() - t1) / a.h2d_steps
        pcie = {"value": reads / dt1, "unit": "reads/s", "ms_per_step": dt1 * 1e3,
                "host_bytes_per_step": int(sum(h.numel() * h.element_size() for h in host.values()))}
        del host

    if world > 1 or a.force_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        # N * (1.001 * W * 1.5 + 12) -- bit planes + one QV byte per base.  Rich reads are priced by the QV
        # planes a threshold makes the pass read (never the 4.5 B/base of planes nobody looks at), a baked set like
        # the QualQV-only case.
        planes = (1 if a.qualqv >= 0 else 0) + (1 if a.rich else 0)
        per_base = 1.5 if (baked or planes <= 1) else 0.5 + planes
        alg_bytes = reads * 1.001 * wl.W * per_base + 12.0 * reads
        assert a.rich or abs(alg_bytes - wl.algorithmic_bytes(n_reads=reads)) < 1.0
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        kernel_name = "pileup_simple_kernel" if a.mode == "simple" else "pileup_kernel"
        out = {
            "metric": "aligned CCS reads/sec (pileup+Fisher, node)",
            "value": reads * world / (dt / a.steps),
            "unit": "reads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 counts, f64 Fisher",
            "data": "synthetic",
            "config": {"workload": "%s: %d reads/GPU x %d columns (HXB2 pol, 1%% minors), QUALQV=%s" % (
                wl.name, reads, W, "unset" if a.qualqv < 0 else a.qualqv) + (", rich QVs, SUBQV=42" if a.rich else ""), "mode": a.mode,
                "window_bounds": "host round trip per step" if step is step_host_bounds else "device-resident",
                "column_slabs": (part.n_slabs if part else (((W + 31) // 32) + 127) // 128),
                "read_layout": "column-slab partition (built at set-up)" if part else "one read set",
                "qv_thresholds": "unset" if a.qualqv < 0 and not a.rich else "applied at pack time (baked)" if baked else "applied in every pass",
                "bases_per_gpu": int(n_bases), "codon_positions": plan.P, "number_of_tests": int(plan.n_tests.item())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None if a.rich else load_traffic(kernel_name, wl.name, reads, a.qualqv, baked),
                         "kernel_ms": kernel_ms, "algorithmic_bytes": alg_bytes},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        print(json.dumps(out), flush=True)
    if world > 1 or a.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
