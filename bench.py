#!/usr/bin/env python3
"""bench.py -- juliet pileup + Fisher hot path on MI355X: aligned CCS reads/s and % of the HBM roofline.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pol_1m] [--reads R] [--scaling weak|strong]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path over one batch of synthetic reads already resident in HBM:

  window bounds of the shard + cleared count matrices + pileup of the shard inside the sample's window (nucleotide
  + codon counts): ONE call, msq_pileup_step -- the stream kernel (the bounds are its side product) + its fold, which
  stores the matrices (--unfused: msq_pileup_begin, then msq_pileup: a bounds / clearing launch of its own)
  -> [N > 1: ONE RCCL group over xGMI -- MAX of the bounds, u32 SUM of the count matrices; the library's own
     communicator, msq_comm_*]
  -> CountNumberOfTests + per-(position, codon) Fisher cells, replicated on every rank.

The window is known from the set-up (as from juliet's --region); the bounds are still recomputed from the reads in
every step and compared with it after the timed loop.  --window device discovers the window on the device in every
step instead (bounds -> [MAX all-reduce] -> pileup reads its window from HBM -> [SUM all-reduce]); --window host
brings it back to the host every step, as MSAByRow's constructor does.

N > 1: every rank holds a shard of one global sample.  "weak" (the line's `value`): `reads` reads per GPU, global
read index = rank * reads + i.  "strong": the workload's reads split over the ranks ([r N / P, (r + 1) N / P)), as
BASELINE config #3 states it; with N > 1 the weak line also carries the strong measurement of the same run as
`strong_scaling`, and --scaling strong makes it the line's `value`.

Set-up (untimed: generating and packing the reads, their column index, the plan, the log-factorial table) ends with
--spinup-ms of untimed steps (default 80 ms): the GPU needs ~50 ms of load before its clocks are up, and 20 steps
behind 5 warm-up steps read 7 % slower than the same steps inside a long run.  `config.clock_spinup` says what ran;
--spinup-ms 0 turns it off.  The W warm-up steps and exactly K timed steps follow.

Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel (the read-streaming pileup kernel) timed with
HIP events on its own stream inside the timed region (every --time-every-th step: an event pair costs the stream
~10 us of bubbles; the median over the timed steps, 25 of them by default); `cpu_baseline` is the oracle (CPU restatement of the reference path) on a bounded sample,
rank 0 at N = 1 only.
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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pol_1m", help="pol_1m (default: 1M x 2843, north-star target), pol_100k, "
                    "hiv_mix_200k, amp20k_10m")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (weak) / of the sample (strong); default: the workload's")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--qualqv", type=int, default=-1, help="QUALQV threshold (env var of the reference); -1 = unset")
    ap.add_argument("--rich", action="store_true", help="reads carry DelQV / SubQV / InsQV as well (4.5 B/base) and the "
                    "reference's default SUBQV=42 applies (src/QvThresholds.cpp:45-56)")
    ap.add_argument("--no-bake", action="store_true", help="with --qualqv: keep the QV plane and apply the threshold "
                    "inside every pileup pass instead of once at set-up (msq_reads_bake_qv)")
    ap.add_argument("--mode", default="auto", choices=["auto", "simple", "stream"])
    ap.add_argument("--layout", default="auto", choices=["auto", "aligned", "normalized"], help="group layout of the resident reads: "
                    "aligned = column space (msq_reads_align at set-up, like the packing: AddRead's placement done once; the pileup "
                    "loads the words it counts), normalized = 32 consecutive bases per group + the group column index (the "
                    "read-space kernel builds the rows in LDS in every pass); auto = aligned where its kernel applies (no QV "
                    "plane read in the pass)")
    ap.add_argument("--no-compare", action="store_true", help="default one-GPU line: do not also time the read-space kernel on the "
                    "normalized layout of the same reads (`read_space_kernel`)")
    ap.add_argument("--window", default="known", choices=["known", "device", "host"], help="see the module docstring")
    ap.add_argument("--host-bounds", action="store_true", help="= --window host")
    ap.add_argument("--unfused", action="store_true", help="msq_pileup_begin + msq_pileup instead of msq_pileup_step")
    ap.add_argument("--no-partition", action="store_true", help="windows wider than 4096 columns: run every slab launch "
                    "over the whole read set instead of over the column-slab partition built at set-up")
    ap.add_argument("--slab-columns", type=int, default=0, help="widest column slab of the partition (multiple of 32, "
                    "<= 4096); narrower than the window: partition even a window that fits one slab")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: create the library's RCCL communicator and "
                    "issue the step's collectives even with one rank")
    ap.add_argument("--rehearse-shared-gpu", action="store_true", help="N > 1 on a box with ONE GPU: every rank uses cuda:0 and the "
                    "exchange runs through the library communicator's shared-memory rehearsal transport (MSQ_COMM_TRANSPORT=shm, "
                    "host copies).  Exercises the launcher, the shards, the msq_comm_* call sites, verification and the JSON "
                    "line of an N > 1 run; its numbers are NOT a measurement "
                    "(the ranks share one GPU) and the line says so")
    ap.add_argument("--exchange", default="lib", choices=["lib", "torch"], help="N > 1: the library's RCCL communicator "
                    "(msq_comm_*) or, for comparison, torch.distributed's RCCL group (also the fallback when the "
                    "library communicator cannot be made; the line's config.exchange says which ran)")
    ap.add_argument("--spinup-ms", type=float, default=80.0, help="set-up ends with untimed steps for this long: the GPU "
                    "needs ~50 ms of load before its clocks are up (20 steps after 5 warm-ups read 7 %% slower than the "
                    "same steps in a long run); 0 = none.  The W warm-up steps and the K timed steps follow as given")
    ap.add_argument("--time-every", type=int, default=4, help="HIP events around the pileup kernel in every n-th step")
    ap.add_argument("--verify", action="store_true", help="after the timed loop: the count matrices of the last step "
                    "against the independent simple kernel's (bit for bit)")
    # data dependence of the pileup kernel: how far the sample strays from its majority sequence
    ap.add_argument("--n-minors", type=int, default=-1, help="minor clones in the sample (default: the workload's 2)")
    ap.add_argument("--minor-frac", type=float, default=-1.0, help="share of the reads in EACH minor clone (default 0.01)")
    ap.add_argument("--minor-codons", type=int, default=-1, help="codons in which a minor clone differs (default 10)")
    ap.add_argument("--sub-rate", type=float, default=-1.0, help="substitution errors per base (default 0.0005244257)")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="reads of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-worker", default="", help=argparse.SUPPRESS)  # name:first:n -> prints the oracle's seconds
    ap.add_argument("--h2d-steps", type=int, default=-1, help="also time K steps that first copy the packed reads from "
                    "pinned host memory (PCIe-inclusive rate, reported beside `value`, never as it); default: 3 on the "
                    "default one-GPU line, 0 elsewhere")
    ap.add_argument("--project-strong", type=int, default=-1, help="one GPU: also time the shard one of this many GPUs would "
                    "hold in a strong-scaling run of the same sample (1/n of the reads, the step's collectives issued on a "
                    "one-rank communicator) and report T(all) / (n T(shard)) as `strong_projection` -- a ceiling: no xGMI "
                    "time is in it; default: 8 on the default one-GPU line, 0 (off) elsewhere")
    ap.add_argument("--launch-dry-run", action="store_true", help="--gpus N > 1 started as a plain `python bench.py`: print "
                    "the command and environment of every rank process the launcher would start, and stop")
    a = ap.parse_args()
    if a.host_bounds:
        a.window = "host"
    # the default one-GPU line carries the PCIe-inclusive rate and the strong-scaling projection beside `value`
    plain = (a.gpus == 1 and not a.force_dist and a.qualqv < 0 and not a.rich and a.mode == "auto" and a.window == "known"
             and not a.unfused and a.reads == 0 and a.workload in ("pol_1m", "pol_100k"))
    if a.h2d_steps < 0:
        a.h2d_steps = 3 if plain else 0
    if a.project_strong < 0:
        a.project_strong = 8 if plain else 0
    a.compare = plain and not a.no_compare and a.layout == "auto"
    return a


def workload_kwargs(a):
    kw = {}
    if a.rich:
        kw["rich_qv"] = True
    if a.n_minors >= 0:
        kw["n_minors"] = a.n_minors
    if a.minor_frac >= 0:
        kw["minor_frac"] = a.minor_frac
    if a.minor_codons >= 0:
        kw["minor_codons"] = a.minor_codons
    if a.sub_rate >= 0:
        kw["sub"] = a.sub_rate
    return kw


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


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_commands(a, argv, port):
    """The rank processes of `python bench.py --gpus N` (N > 1, no launcher around it): one fresh interpreter per GPU,
    the environment torch.distributed.run would give it, rendezvous on loopback."""
    argv = [x for x in argv if x != "--launch-dry-run"]
    cmds = []
    for r in range(a.gpus):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(a.gpus), "LOCAL_WORLD_SIZE": str(a.gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "GLOO_SOCKET_IFNAME": "lo",
               "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               "MSQ_BENCH_LAUNCHED": "1"}  # the parent holds the CPU baseline: rank 0 does not time it again
        cmds.append({"rank": r, "argv": [sys.executable, "-u", os.path.abspath(__file__)] + argv, "env": env})
    return cmds


def launch(a):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: this process never touches the GPU (no torch import,
    no HIP library) -- it times the CPU baseline, starts N rank processes, relays rank 0's JSON line with
    `cpu_baseline` merged in, and fails when any rank fails."""
    import subprocess
    cmds = rank_commands(a, sys.argv[1:], free_port())
    if a.launch_dry_run:
        print(json.dumps({"launcher": "bench.py", "ranks": cmds}))
        return 0
    cpu = None
    if a.cpu_sample > 0:
        from minorseq_amd import synth as _synth
        cpu = cpu_baseline(a.workload, min(a.cpu_sample, a.reads or _synth.workload(a.workload).n_reads))
    procs = []
    for c in cmds:
        env = dict(os.environ)
        env.update(c["env"])
        # rank 0's stdout carries the line; the other ranks print nothing there, and whatever they do print goes to stderr
        procs.append(subprocess.Popen(c["argv"], env=env, stdout=subprocess.PIPE if c["rank"] == 0 else sys.stderr,
                                      text=True if c["rank"] == 0 else None))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    live = set(range(len(procs)))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = code if code > 0 else 128 - code
                print("bench: rank %d exited with %d" % (r, code), file=sys.stderr)
                break
        time.sleep(0.05)
    if rc:  # one rank failed: the others would wait for it in a collective for ever -- end exactly the ones started here
        for r in live:
            procs[r].terminate()
        t_end = time.time() + 10.0
        for r in live:
            try:
                procs[r].wait(max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
        return rc
    reader.join(10.0)
    lines = [ln for ln in "".join(out0).splitlines() if ln.startswith("{")]
    if not lines:
        print("bench: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    out = json.loads(lines[-1])
    if cpu is not None:
        out["cpu_baseline"] = cpu
    out.setdefault("config", {})["launcher"] = "bench.py started %d rank processes itself (no WORLD_SIZE in its environment)" % a.gpus
    print(json.dumps(out))
    return 0


class Run:
    """One shard resident in HBM with everything a step needs, and the timed loop over it."""

    def __init__(self, a, ctx, ex, wl, first, reads, global_reads, window, layout=None):
        import torch
        from minorseq_amd import abi, device
        self.a, self.ctx, self.ex, self.wl, self.reads = a, ctx, ex, wl, reads
        self.torch, self.abi, self.device = torch, abi, device
        L = self.L = abi.lib()
        self.thr = abi.QvThresholds.make(qual=a.qualqv, sub=42 if a.rich else -1)
        self.mode = {"auto": abi.MSQ_PILEUP_AUTO, "simple": abi.MSQ_PILEUP_SIMPLE, "stream": abi.MSQ_PILEUP_STREAM}[a.mode]
        # ---- setup (untimed): reads into HBM, plan, log-factorial table
        dr = device.synth_device(ctx, wl, first, reads)
        self.n_bases = dr.n_groups * 32
        # a run's QV thresholds are fixed before the reads are loaded: they are applied once, at pack time (untimed like
        # the packing itself), and the passes below stream the baked set -- as the front does (MSAByRow::Finish)
        self.baked = (a.qualqv >= 0 or a.rich) and not a.no_bake and a.h2d_steps == 0
        if self.baked:
            src = dr
            dr = device.bake_qv(ctx, src, self.thr)
            ctx.sync()
            del src
        if window is None:
            window = ex.allreduce_bounds(*device.window_bounds(ctx, dr))
        self.b0, self.e0 = window
        b0, W = self.b0, self.e0 - self.b0
        self.W = W
        # the column-aligned layout (set-up, like the packing and the baking): on the word grid of the sample's window
        unbaked_planes = (not self.baked) and (a.qualqv >= 0 or a.rich)
        layout = a.layout if layout is None else layout
        self.aligned = layout == "aligned" or (layout == "auto" and not unbaked_planes and a.mode == "auto" and
                                               a.window != "device" and not a.slab_columns and not a.no_partition)
        self.align_ms = None
        if self.aligned:
            ctx.sync()
            t_al = time.perf_counter()
            al = device.align(ctx, dr, self.thr, b0)
            ctx.sync()
            self.align_ms = (time.perf_counter() - t_al) * 1e3
            self.n_bases = al.n_groups * 32
            dr = al
        self.dr = dr
        gb, ge = wl.gene_arrays()
        self.plan = device.Plan(ctx, gb, ge, wl.reference if wl.use_reference else "", b0 + 1, W)
        ctx.lfact_reserve(2 * global_reads + 2)
        self.counts = torch.zeros(W * 70, dtype=torch.int32, device="cuda")  # [W][6] nucleotide then [W][64] codon, one buffer
        self.nuc, self.codon = self.counts[:W * 6], self.counts[W * 6:]
        self.em = abi.ErrorModel()
        abi.check(L.msq_error_model_default(C.byref(self.em)))
        self.bounds = torch.zeros(2, dtype=torch.int32, device="cuda")  # {-begin0, end0}, device-resident
        # windows wider than one slab of the streaming kernel: the reads are laid out per column slab at set-up (pack
        # time, untimed like the packing itself), so that every slab launch streams only its share of the reads
        self.part = None
        if (W > 4096 or 0 < a.slab_columns < W) and not a.no_partition and a.mode != "simple" and not self.aligned:
            self.part = device.Partition(ctx, dr, b0, W, a.slab_columns)
        self.launches_per_step = 1  # timing records per step: counted before the timed steps (run)
        self.flow = "host" if (a.window == "host" or (a.mode == "simple" and a.window == "device")) else a.window
        if self.part and self.flow == "device":
            self.flow = "known"  # a partition is laid out for its window

    def cells(self):
        p = self.plan
        self.abi.check(self.L.msq_call_cells(self.ctx.h, p.h, self.codon.data_ptr(), C.byref(self.em), 100.0, p.positions.data_ptr(),
                                             p.p.data_ptr(), p.expected.data_ptr(), p.n_tests.data_ptr()))

    def step(self):
        L, abi, ctx, dr, ex = self.L, self.abi, self.ctx, self.dr, self.ex
        if self.flow == "known":
            # bounds of the shard + cleared matrices in one launch; the pileup inside the sample's window; one RCCL group
            if self.part or self.mode != abi.MSQ_PILEUP_AUTO or self.a.unfused:
                abi.check(L.msq_pileup_begin(ctx.h, dr.view, self.bounds.data_ptr(), self.counts.data_ptr(), self.counts.numel()))
                if self.part:
                    abi.check(L.msq_pileup_partition(ctx.h, self.part._h, self.thr, self.nuc.data_ptr(), self.codon.data_ptr()))
                else:
                    abi.check(L.msq_pileup(ctx.h, dr.view, self.thr, self.b0, self.W, self.nuc.data_ptr(), self.codon.data_ptr(), self.mode))
            else:  # the same as one call: the bounds come out of the pileup kernel, its fold stores the matrices
                abi.check(L.msq_pileup_step(ctx.h, dr.view, self.thr, self.b0, self.W, self.bounds.data_ptr(), self.counts.data_ptr()))
            ex.allreduce_step(self.bounds, self.counts)
        elif self.flow == "device":  # the window discovered on the device: nothing in the step waits for the host
            abi.check(L.msq_window_bounds_device(ctx.h, dr.view, self.bounds.data_ptr()))
            ex.allreduce_bounds_device(self.bounds)  # N > 1: one 2-int MAX
            self.counts.zero_()
            abi.check(L.msq_pileup_device_window(ctx.h, dr.view, self.thr, self.bounds.data_ptr(), self.b0, self.W, self.nuc.data_ptr(),
                                                 self.codon.data_ptr()))
            ex.allreduce_counts(self.counts)
        else:  # the window comes back to the host every step, as MSAByRow's constructor needs it
            w = ex.allreduce_bounds(*self.device.window_bounds(ctx, dr))  # MSAByRow pass 1 (UpdateBoundaries)
            self.bounds.copy_(self.torch.tensor([-w[0], w[1]], dtype=self.torch.int32))
            self.counts.zero_()
            if self.part:
                abi.check(L.msq_pileup_partition(ctx.h, self.part._h, self.thr, self.nuc.data_ptr(), self.codon.data_ptr()))
            else:
                abi.check(L.msq_pileup(ctx.h, dr.view, self.thr, self.b0, self.W, self.nuc.data_ptr(), self.codon.data_ptr(), self.mode))
            ex.allreduce_counts(self.counts)
        self.cells()

    def timed(self, barrier, agree=lambda x: x):
        a, abi, L, ctx = self.a, self.abi, self.L, self.ctx
        self.spinup_steps = 0
        if a.spinup_ms > 0:  # clocks up (set-up, like packing the reads).  A step holds collectives: every rank runs the
            t_up = time.perf_counter()  # same number of steps -- eight to see how long one takes, then what the ranks agree on
            for _ in range(8):
                self.step()
            ctx.sync()
            t8 = max(time.perf_counter() - t_up, 1e-6)
            more = int(agree(min(20000.0, max(0.0, a.spinup_ms * 1e-3 / t8 * 8.0 - 8.0))))
            for _ in range(more):
                self.step()
            ctx.sync()
            self.spinup_steps = 8 + more
        for _ in range(a.warmup):
            self.step()
        abi.check(L.msq_pileup_check(ctx.h))
        abi.check(L.msq_call_check(ctx.h))
        # timing records per step: one per pileup call that reaches the kernel -- a partition's slabs take one launch
        # together where the library can merge them, else one each; counted on one more untimed step
        abi.check(L.msq_timing_enable(ctx.h, 256))
        self.step()
        probe, n_probe = (C.c_float * 256)(), C.c_int32(0)
        abi.check(L.msq_timing_read(ctx.h, probe, 256, C.byref(n_probe)))
        abi.check(L.msq_timing_enable(ctx.h, 0))
        self.launches_per_step = max(1, n_probe.value)
        stride = 1 if self.launches_per_step > 1 else max(1, min(a.time_every, a.steps))
        n_slots = (a.steps * self.launches_per_step + stride - 1) // stride
        abi.check(L.msq_timing_enable_strided(ctx.h, n_slots, stride))
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            self.step()
        barrier()
        dt = time.perf_counter() - t0
        abi.check(L.msq_pileup_check(ctx.h))
        abi.check(L.msq_call_check(ctx.h))
        if self.bounds.tolist() != [-self.b0, self.e0]:
            raise SystemExit("window bounds of the timed steps %s differ from the set-up window [%d, %d)" % (self.bounds.tolist(), self.b0, self.e0))
        ms = (C.c_float * n_slots)()
        n_ms = C.c_int32(0)
        path_ms = None
        def median(v):
            v = sorted(v)
            return 0.0 if not v else v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
        if self.launches_per_step == 1:
            abi.check(L.msq_timing_read_path(ctx.h, ms, n_slots, C.byref(n_ms)))
            path_ms = median(ms[:n_ms.value])
        abi.check(L.msq_timing_read(ctx.h, ms, n_slots, C.byref(n_ms)))
        abi.check(L.msq_timing_enable(ctx.h, 0))
        k = self.launches_per_step
        timed_steps = max(1, n_ms.value // k)
        # per timed step: the slab launches of a partition added up; the median over the timed steps
        kernel_ms = median([sum(ms[i * k:(i + 1) * k]) for i in range(n_ms.value // k)])
        return dt, kernel_ms, path_ms, timed_steps

    def verify(self):
        """The count matrices the last step left against the simple kernel's on the same shard (+ the same exchange)."""
        torch, abi = self.torch, self.abi
        ref = torch.zeros_like(self.counts)
        W = self.W
        abi.check(self.L.msq_pileup(self.ctx.h, self.dr.view, self.thr, self.b0, W, ref[:W * 6].data_ptr(), ref[W * 6:].data_ptr(),
                                    abi.MSQ_PILEUP_SIMPLE))
        abi.check(self.L.msq_pileup_check(self.ctx.h))
        self.ex.allreduce_counts(ref)
        if not torch.equal(ref, self.counts):
            bad = int((ref != self.counts).sum())
            raise SystemExit("--verify: %d of %d count-matrix entries differ from the simple kernel's" % (bad, ref.numel()))
        return True

    def deviation_share(self):
        """Share of the piled-up ACGT bases that differ from their column's majority base: what the stream kernel
        sends down its per-base atomic path instead of its bit-sliced counters."""
        n = self.nuc.view(-1, 6)[:, :4].to(self.torch.int64)
        tot = int(n.sum())
        return 1.0 - int(n.max(1).values.sum()) / max(1, tot)


def main():
    a = parse()
    if a.cpu_worker:
        name, first, n = a.cpu_worker.split(":")
        print(_cpu_worker((name, int(first), int(n))))
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:  # a plain `python bench.py --gpus N`: start the ranks ourselves
        sys.exit(launch(a))
    if a.launch_dry_run:
        raise SystemExit("--launch-dry-run describes the launcher of a plain `python bench.py --gpus N` with N > 1")
    if os.environ.get("MSQ_BENCH_WATCHDOG"):  # debugging aid: every thread's stack to stderr, then exit, after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["MSQ_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus %d` (it starts its own ranks) or "
                         "under torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus, a.gpus))

    def cpu_now():
        from minorseq_amd import synth as _synth
        return cpu_baseline(a.workload, min(a.cpu_sample, a.reads or _synth.workload(a.workload).n_reads))
    # the CPU baseline is timed before this job touches the GPU: here at N = 1; by the launcher when bench.py started
    # the ranks itself; by rank 0 behind the rendezvous (the other ranks wait in a barrier) under torch.distributed.run
    want_cpu = a.cpu_sample > 0 and rank == 0 and os.environ.get("MSQ_BENCH_LAUNCHED") != "1"
    cpu = cpu_now() if (want_cpu and world == 1) else None
    # stdout carries the ONE JSON line and nothing else: whatever libraries print while the run is set up (RCCL's
    # version banner, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from minorseq_amd import abi, device, sharding, synth

    n_dev = torch.cuda.device_count()  # counting does not initialise the GPU
    shared = bool(a.rehearse_shared_gpu) and world > 1
    if shared:
        if n_dev < 1:
            raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
        local = 0  # every rank on the box's one GPU
    elif n_dev <= local or n_dev < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        raise SystemExit("bench.py --gpus %d needs %d GPUs, found %d" % (a.gpus, a.gpus, n_dev))
    multi = world > 1 or a.force_dist
    if world > 1:
        # control plane only (the communicator id, the barrier, the max-over-ranks clock): gloo over loopback; the
        # data path is the library's RCCL communicator on the context's stream
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if want_cpu:
            cpu = cpu_now()
        dist.barrier()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    ctx = device.Ctx(local)
    ex_note = None
    if shared:
        # the library's own communicator and entry points (msq_comm_*), over its shared-memory rehearsal transport: RCCL
        # refuses ranks that share a device
        os.environ["MSQ_COMM_TRANSPORT"] = "shm"
        ex = sharding.LibExchange(ctx, rank, world, sharding.torch_bootstrap)
        ex_note = ("REHEARSAL, not a measurement: %d ranks share cuda:0; bounds MAX + counts SUM through the library communicator "
                   "(msq_comm_*) over its shared-memory rehearsal transport (host copies); RCCL and xGMI are not in this run" % world)
        print("bench: " + ex_note, file=sys.stderr)
    elif world > 1:
        # every rank must take the same road.  msq_comm_create is a collective (ncclCommInitRank): a rank that cannot
        # even load RCCL must say so BEFORE the others enter it, so the ranks first agree on a local, non-collective
        # check (msq_comm_available) and only then create the communicator; if any of them cannot, all of them take
        # torch.distributed's RCCL group (same collectives, same stream), and the line says so
        err = "--exchange torch" if a.exchange == "torch" else ""
        if not err and abi.lib().msq_comm_available() != abi.MSQ_OK:
            err = abi.lib().msq_last_error().decode("utf-8", "replace")
        bad = [None] * world
        dist.all_gather_object(bad, err)
        ex = None
        if not any(bad):
            try:
                ex = sharding.LibExchange(ctx, rank, world, sharding.torch_bootstrap)
                err = ""
            except Exception as e:  # noqa: BLE001 -- reported in the JSON line
                ex, err = None, "%s: %s" % (type(e).__name__, e)
            dist.all_gather_object(bad, err)
        if any(bad):
            if ex is not None:
                ex.close()
            grp = dist.new_group(backend="nccl")

            class _CudaTorchExchange(sharding.TorchExchange):
                def allreduce_bounds(self, begin0, end0):
                    return sharding.TorchExchange.allreduce_bounds(self, begin0, end0, device=torch.device("cuda", local))
            ex = _CudaTorchExchange(grp)
            ex_note = "torch.distributed RCCL group over %d ranks (the library communicator was not used: %s)" % (world, next(b for b in bad if b))
            print("bench: " + ex_note, file=sys.stderr)
    elif a.force_dist:  # rehearsal on one GPU: a one-rank communicator, no process group needed
        ex = sharding.LibExchange(ctx, 0, 1, lambda raw: raw)
    else:
        ex = sharding.NoExchange()
    comm_ranks = abi.lib().msq_comm_world(ctx.h) if isinstance(ex, sharding.LibExchange) else None

    wl = synth.workload(a.workload, **workload_kwargs(a))
    reads = a.reads or wl.n_reads

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def measure(scaling):
        if scaling == "weak":
            first, n, total = rank * reads, reads, reads * world
        else:
            first, n = sharding.shard_range(reads, rank, world)
            total = reads
        run = Run(a, ctx, ex, wl, first, n, total, None)
        dt, kernel_ms, path_ms, timed_steps = run.timed(barrier, max_over_ranks)
        dt = max_over_ranks(dt)
        verified = run.verify() if a.verify else None
        return run, total, dt, kernel_ms, path_ms, timed_steps, verified

    primary = measure(a.scaling)
    run, total, dt, kernel_ms, path_ms, timed_steps, verified = primary
    secondary = None

    class _RunFacts:  # what the JSON line says about the primary run, kept when the run itself makes room for the next
        def __init__(self, r):
            self.aligned, self.align_ms = r.aligned, getattr(r, "align_ms", 0.0)
            self.has_gcol, self.hap_count = bool(r.dr.view.gcol), int(r.dr.view.hap_count)
            self.launches_per_step, self.spinup_steps = r.launches_per_step, r.spinup_steps
    facts = _RunFacts(run)
    if world > 1 and a.scaling == "weak":  # the strong-scaling measurement of the same run, carried along
        keep = (run.n_bases, run.W, run.plan.P, int(run.plan.n_tests.item()), run.baked, run.flow, run.part.n_slabs if run.part else None,
                run.deviation_share(), run.reads)
        del run, primary
        torch.cuda.empty_cache()
        s_run, s_total, s_dt, s_kernel_ms, s_path_ms, _, _ = measure("strong")
        secondary = {"value": s_total / (s_dt / a.steps), "unit": "reads/s", "ms_per_step": s_dt / a.steps * 1e3,
                     "reads_per_gpu": s_run.reads, "kernel_ms": s_kernel_ms, "path_ms": s_path_ms,
                     "sample": "%d reads of %s split over %d GPUs" % (s_total, wl.name, world)}
        n_bases, W, P, n_tests, baked, flow, n_slabs, dev_share, shard_reads = keep
    else:
        n_bases, W, P, n_tests, baked, flow, n_slabs, dev_share, shard_reads = (
            run.n_bases, run.W, run.plan.P, int(run.plan.n_tests.item()), run.baked, run.flow, run.part.n_slabs if run.part else None,
            run.deviation_share(), run.reads)

    pcie = None
    if a.h2d_steps > 0 and world == 1:  # the boundary handing over host buffers: packed reads cross PCIe every step
        dr = run.dr
        if run.aligned:
            # the aligned set as the host would hold it: tables, the words, the guide (3 KB); nothing derived is rebuilt
            v = dr.view
            parts = [(v.ref_start, reads * 4), (v.ref_end, reads * 4), (v.group_off, (reads + 1) * 8), (v.bop, dr.n_groups * 16),
                     (v.guide, int(v.guide_words) * 32)]
            host = []
            for addr, nbytes in parts:
                h = torch.empty(max(nbytes, 1), dtype=torch.uint8).pin_memory()
                run.abi.check(run.L.msq_memcpy_d2h(ctx.h, h.data_ptr(), addr, nbytes))
                host.append((addr, h, nbytes))
            barrier()
            t1 = time.perf_counter()
            for _ in range(a.h2d_steps):
                for addr, h, nbytes in host:
                    run.abi.check(run.L.msq_memcpy_h2d(ctx.h, addr, h.data_ptr(), nbytes))
                run.step()
            barrier()
            dt1 = (time.perf_counter() - t1) / a.h2d_steps
            host_bytes = int(sum(nb for _, _, nb in host))
        else:
            # what crosses: the tables, the bit planes and the QV planes this run's thresholds read -- not the planes
            # nobody looks at, and not the column index, which is derived data: rebuilt on the device after every copy
            need = {"ref_start", "ref_end", "group_off", "bop"} | ({"qual"} if a.qualqv >= 0 else set()) | ({"subqv"} if a.rich else set())
            host = {f: t.cpu().pin_memory() for f, t in dr.t.items() if t is not None and f in need}
            barrier()
            t1 = time.perf_counter()
            for _ in range(a.h2d_steps):
                for f, h in host.items():
                    dr.t[f].copy_(h, non_blocking=True)
                if dr.t.get("gcol") is not None:
                    run.abi.check(run.L.msq_reads_index(ctx.h, dr.view, dr.t["gcol"].data_ptr()))
                run.step()
            barrier()
            dt1 = (time.perf_counter() - t1) / a.h2d_steps
            host_bytes = int(sum(h.numel() * h.element_size() for h in host.values()))
        pcie = {"value": reads / dt1, "unit": "reads/s", "ms_per_step": dt1 * 1e3, "host_bytes_per_step": host_bytes}
        del host

    # the same reads in the normalized layout through the read-space kernel (the dominant kernel of rounds 1 and 2), same run
    compare = None
    if a.compare and world == 1 and run.aligned:
        try:
            run0 = Run(a, ctx, ex, wl, 0, reads, reads, (run.b0, run.e0), layout="normalized")
            dt0, k0_ms, p0_ms, _ = run0.timed(barrier)
            compare = {"layout": "normalized groups + group column index (0.625 B/base streamed); rows built in LDS in every pass",
                       "ms_per_step": dt0 / a.steps * 1e3, "value": reads / (dt0 / a.steps), "unit": "reads/s", "kernel_ms": k0_ms, "path_ms": p0_ms}
            del run0
        except Exception as e:  # noqa: BLE001 -- the line says why the comparison is missing
            compare = {"error": "%s: %s" % (type(e).__name__, e)}

    projection = None
    if a.project_strong > 1 and world == 1:
        # what a strong-scaling run over n GPUs can reach at best: one rank's shard (1/n of the sample's reads, inside the
        # sample's window, the log-factorial table of the whole sample) with the step's collectives issued on a
        # one-rank communicator -- RCCL's launch is in it, xGMI time is not
        n = a.project_strong
        ex1 = None
        try:
            ex1 = ex if isinstance(ex, sharding.LibExchange) else sharding.LibExchange(ctx, 0, 1, lambda raw: raw)
            first1, n1 = sharding.shard_range(reads, 0, n)
            run1 = Run(a, ctx, ex1, wl, first1, n1, reads, (run.b0, run.e0))
            dt1s, k1_ms, p1_ms, _ = run1.timed(barrier)
            t_all, t_shard = dt / a.steps, dt1s / a.steps
            projection = {"ranks": n, "reads_per_gpu": n1, "ms_per_step": t_shard * 1e3, "kernel_ms": k1_ms, "path_ms": p1_ms,
                          "efficiency_ceiling": t_all / (n * t_shard),
                          "note": "derived, not measured on %d GPUs: T(%d reads) / (%d x T(%d reads)) on ONE GPU, the shard's step with "
                                  "its RCCL group issued on a one-rank communicator; xGMI time is not in it" % (n, reads, n, n1)}
            del run1
        except Exception as e:  # noqa: BLE001 -- the line says why the projection is missing
            projection = {"ranks": n, "error": "%s: %s" % (type(e).__name__, e)}
        finally:
            if ex1 is not None and ex1 is not ex:
                ex1.close()

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        # N * (1.001 * W * 1.5 + 12) -- bit planes + one QV byte per base.  Rich reads are priced by the QV
        # planes a threshold makes the pass read (never the 4.5 B/base of planes nobody looks at), a baked set like
        # the QualQV-only case.
        planes = (1 if a.qualqv >= 0 else 0) + (1 if a.rich else 0)
        per_base = 1.5 if (baked or planes <= 1) else 0.5 + planes
        alg_bytes = shard_reads * 1.001 * wl.W * per_base + 12.0 * shard_reads
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # what the kernel really streams: 16 B of bit planes per 32-base group, 32 B per group for every QV plane a
        # threshold makes it read, 12 B of tables per read (the QV planes of an unmasked or baked set are never touched)
        # + 4 B per group of column index when the set carries one and the kernel variant uses it (no QV planes to read)
        aligned = facts.aligned
        indexed = (not aligned) and facts.has_gcol and (baked or planes == 0) and a.mode != "simple"
        groups = facts.hap_count if aligned else 0  # haplotype groups: + 4 B/read of the group-ordered read list
        per_read = (24.0 if groups >= 2 else 20.0) if aligned else 12.0
        real_bytes = n_bases / 32 * (16 + 32 * (0 if (baked or aligned) else planes) + (4 if indexed else 0)) + per_read * shard_reads
        kernel_name = "pileup_simple_kernel" if a.mode == "simple" else "pileup_rows_kernel" if aligned else "pileup_kernel"
        traffic = None if (a.rich or a.scaling == "strong" and world > 1) else load_traffic(kernel_name, wl.name, shard_reads, a.qualqv, baked)
        physical = (traffic or real_bytes) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        limiter = "valu/latency" if physical < 0.8 * 6300.0 else "hbm"
        # a leaner format than 1.5 B/base may exceed frac 1 only by reading less -- state the format
        qv_read = 0 if baked else planes
        fmt = ("column-aligned groups (msq_reads_align at set-up: a group = a 32-column word of its read's row, 2-bit nucleotide + 2-bit op "
               "per column, inserted bases left out): 0.5 B per column-consuming base + %d B/read of tables%s; no QV plane, no column index "
               "read; algorithmic bytes price 1.5 B/base as SURVEY 8d states them"
               % (per_read, " and read list (%d haplotype groups, each counted against its own guide)" % groups if groups >= 2 else "")) if aligned else \
              "0.5 B/base bit planes (2-bit nucleotide + 2-bit op, sliced per 32-base group)" + \
              (" + 0.125 B/base group column index" if indexed else "") + " + 12 B/read; " + \
              ("%d QV plane%s read in the pass (1 B/base each)" % (qv_read, "" if qv_read == 1 else "s") if qv_read else
               "QV planes not read: " + ("thresholds applied at pack time" if baked else "QUALQV unset, no threshold can mask a base")) + \
              "; algorithmic bytes price 1.5 B/base as SURVEY 8d states them"
        out = {
            "metric": "aligned CCS reads/sec (pileup+Fisher, node)",
            "value": total / (dt / a.steps),
            "unit": "reads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "u32 counts, f64 Fisher",
            "data": "synthetic",
            "config": {"workload": "%s: %d reads/GPU x %d columns (%d minor clones at %g %% each in %d codons, %g substitutions/base), QUALQV=%s" % (
                wl.name, shard_reads, W, wl.params.n_minors, 100.0 * wl.params.thr_minor / 4294967296.0,
                len([s for s in wl.minor_sites if s[0] == 1]), wl.params.thr_sub / 4294967296.0,
                "unset" if a.qualqv < 0 else a.qualqv) + (", rich QVs, SUBQV=42" if a.rich else ""), "mode": a.mode,
                "window": {"known": "known at set-up; bounds recomputed per step (inside the pileup kernel, msq_pileup_step; --unfused: in a launch of their own) and checked",
                           "device": "discovered on the device per step", "host": "host round trip per step"}[flow],
                "exchange": (ex_note or (("one RCCL group per step (bounds MAX + counts SUM)" if flow == "known" else "two RCCL all-reduces per step")
                                          + ", library communicator (msq_comm_*) of %d rank%s as RCCL reports it" % (comm_ranks, "" if comm_ranks == 1 else "s")))
                            if multi else "none (one GPU)",
                "column_slabs": (n_slabs if n_slabs else (((W + 31) // 32) + 125) // 126 if facts.aligned else (((W + 31) // 32) + 127) // 128),
                "pileup_launches_per_step": facts.launches_per_step,
                "read_layout": ("column-slab partition (built at set-up)" if n_slabs else
                                "one read set, column-aligned (msq_reads_align at set-up: %.1f ms; %s)" % (facts.align_ms, "%d haplotype groups" % groups if groups >= 2 else "one population") if aligned else "one read set, normalized groups"),
                "qv_thresholds": "unset" if a.qualqv < 0 and not a.rich else "applied at pack time (baked; the bake pass is set-up, not timed)" if baked else "applied in every pass",
                "clock_spinup": ("%d untimed steps (%.0f ms) at the end of set-up, before the warm-up steps" % (facts.spinup_steps, a.spinup_ms)
                                 if facts.spinup_steps else "none"),
                "bases_per_gpu": int(n_bases), "codon_positions": P, "number_of_tests": n_tests,
                "bases_off_majority": dev_share},
            # `bound` says what limits the kernel ("hbm" from 80 % of what a device copy reaches on this part, 6.3 TB/s,
            # i.e. 5.0 TB/s of real traffic; "valu/latency" below); achieved / peak / frac are priced against the HBM
            # roof either way (`roof`), with SURVEY 8d's algorithmic bytes
            "roofline": {"bound": limiter, "roof": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("profiles/traffic.json (committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this kernel, workload "
                                            "and size; not measured in this run)" if traffic else None),
                         "format": fmt,
                         "kernel_ms": kernel_ms, "algorithmic_bytes": alg_bytes, "timed_steps": timed_steps,
                         # the physical side of the same launch: bytes really moved (PMC traffic when a pass of this
                         # exact configuration is committed, else the bytes the kernel streams) over the same time
                         "streamed_bytes": real_bytes, "column_index": indexed, "physical_gbs": physical, "frac_physical": physical / HBM_PEAK_GBS,
                         "limiter": limiter,
                         # K1 + fold [+ exchange] + K2, SURVEY 8d's t_kernel(K1 + K2)
                         "path_ms": path_ms, "frac_path": (alg_bytes / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if path_ms else None},
        }
        if secondary is not None:
            out["strong_scaling"] = secondary
        if verified is not None:
            out["verified_against_simple_kernel"] = verified
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if projection is not None:
            out["strong_projection"] = projection
        if compare is not None:
            out["read_space_kernel"] = compare
        if shared:
            out["rehearsal"] = "ranks share one GPU, exchange over host shared memory: control flow only, `value` is not a measurement"
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ex.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
