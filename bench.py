#!/usr/bin/env python3
"""bench.py — ZMWs/s of the CCS per-ZMW consensus hot path on N MI355X (BASELINE.json metric).

A "step" = one pass of the whole hot path (tables, POA draft, subread->draft alignment, windowing, candidate filter,
Arrow polish + QVs, stitch) over one batch of synthetic ZMWs.  Workload at N=1 is BASELINE.json configs[1]: 10 passes x
10 kb synthetic subreads, 16384 ZMWs per step (four k_poa_dp waves per SIMD instead of two: +2 % over 8192-ZMW steps on the same box, 32768 gives no more —
DESIGN.md 4, round 4); successive steps take successive DISTINCT batches (7 distinct batches = 115 k ZMWs, the configs[1] job; a longer run cycles through
them).  ZMWs shard across ranks with no collective on the data
path (weak scaling: per-GPU batch fixed).

Timed region: from the first submit to the last result of K steps through the library's asynchronous
boundary (ccsx_submit / ccsx_wait): batches start in page-locked HOST memory, go H2D, through every kernel, and their
results come back D2H into page-locked host memory; up to three batches are in flight, so the copies of batch k+1 / k-1 and
the draft stage of batch k+1 run under the polish stage of batch k (two compute streams; `--serial-stages` = one).  `value`
is that PCIe-inclusive rate.  `resident_zmws_per_s` is the same work counted over the kernels alone (HIP events: first kernel
of the first timed batch to the last kernel of the last): the rate with inputs already in HBM.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, its duration measured live with HIP events on the
stream it is launched on (ccsx_ticket_timings); with `--pmc` (run under `rocprofv3 --pmc ...`) nothing but the headline steps
runs, so a counter pass profiles exactly this workload and `tools/mk_traffic.py` turns the passes into profiles/rNN_traffic.json,
whose `head` field ties it to a commit; `cpu_baseline` times the CPU restatement (oracle, kind "port") on the host cores over a
bounded sample of the same workload, and the reference `ccs` binary is probed for (command -v ccs) and, when present, RUN on
the same synthetic subreads (timing + identity against the truth: `reference_ccs`).  `extra` carries the other BASELINE shapes
(c1 / c4 / c5) through the same pipeline, a few steps each.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TRAFFIC_FILE = "r06_traffic.json"       # this round's committed PMC passes (tools/prof_round.sh -> tools/mk_traffic.py)
TRAFFIC_FILES_EXTRA = {"c4": "r06_traffic_c4.json", "c5": "r06_traffic_c5.json"}   # the same passes for the other BASELINE shapes (extra.<shape>.roofline.traffic)
QVCAL_FILE = "r06_qv_calibration.json"  # predicted vs empirical accuracy per rq bin (tools/qv_calibration.py)
VALU_PEAK_LANE_OPS = 78.6e12            # non-packed VALU issue of one MI355X: 1024 SIMDs x 32 lanes per cycle (v_add / v_mul_f32 issue a wave64
                                        # in 2 cycles, profiles/r02_valu_peak.txt) x 2.4 GHz; packed fp32 (157 TFLOP/s with FMA) is not what a DP cell can use
NOMINAL_OPS_PER_CELL = 8                # "8 flop/cell nominal"


def gpu_clocks(device: int) -> dict:
    """sclk / mclk / power of the device as rocm-smi reports them right now (GPU boxes of the pool differ by up to
    35 % on the VALU-bound kernels; the bench line carries what the box ran at so that a spread can be attributed)"""
    out = {}
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--json"],
                           capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "socclk", "power", "performance level")):
                out[k] = v
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"[:120]
    return out


def csrc_sha16() -> str | None:
    """hash of the kernel / library sources (ccs_amd/csrc/*, include/ccsx.h): what ties a counter file to a build where there is no git (the GPU box)"""
    import hashlib
    try:
        h = hashlib.sha256()
        d = os.path.join(ROOT, "ccs_amd", "csrc")
        for f in sorted(os.listdir(d)) + ["../../include/ccsx.h"]:
            if f.endswith((".hip", ".cpp", ".h")):
                h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
        return h.hexdigest()[:16]
    except OSError:
        return None


def kernels_changed_since(rev: str | None) -> bool | None:
    """True / False when git can tell whether ccs_amd/csrc differs between `rev` and the working tree, None without a repository"""
    if not rev:
        return None
    try:
        # (outside a repository `git diff` silently becomes `git diff --no-index REV PATH` and reports a difference: VERDICT r04 — ask for the repository first)
        top = subprocess.run(["git", "-C", ROOT, "rev-parse", "--show-toplevel"], capture_output=True, text=True, timeout=20)
        if top.returncode != 0 or os.path.realpath(top.stdout.strip()) != os.path.realpath(ROOT):
            return None
        r = subprocess.run(["git", "-C", ROOT, "diff", "--quiet", rev.split("+")[0], "--", "ccs_amd/csrc"], capture_output=True, timeout=20)
        return None if r.returncode not in (0, 1) else bool(r.returncode)
    except Exception:
        return None


def effective_cores() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(int(q) / int(p)))))
    except Exception:
        pass
    return n


def host_memory_budget() -> int:
    """bytes of host memory this process may pin: MemAvailable capped by the cgroup limit, halved for safety"""
    avail = 1 << 62
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except Exception:
        pass
    try:
        m = open("/sys/fs/cgroup/memory.max").read().strip()
        if m != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read().strip())
            avail = min(avail, int(m) - cur)
    except Exception:
        pass
    return max(0, avail // 2)


def git_head() -> str | None:
    """the commit this tree is at: from git where there is a repository, else the stamp __graft_entry__.build() left beside the objects
    (the GPU box gets a snapshot without .git)"""
    try:
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        if h:
            return h
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, "ccs_amd", "build", "HEAD")).read().strip() or None
    except OSError:
        return None


# BASELINE.json configs (SURVEY.md §8 sizes): (passes, template length, ZMWs per GPU per step in the default run).  Every preset keeps
# at least 8192 POA graphs resident where memory allows (one wave per graph: fewer leave SIMDs idle in the draft stage); the headline
# shape takes 16384 per step (k_poa_dp: four waves per SIMD; 34.8 k against 34.0 k ZMWs/s at 8192 on the same box, 32768: 34.7 k).
# `fit_zmws` halves the batch until its page-locked copies fit the host.
WORKLOADS = {"c1": (3, 1000, 65536), "c2": (10, 10000, 16384), "c4": (30, 20000, 8192), "c5": ((3, 50), (1000, 25000), 8192)}


def _span(v):
    a = [int(x) for x in str(v).split("-")]
    return a[0] if len(a) == 1 else (a[0], a[1])


def _mean(v):
    return (v[0] + v[1]) / 2 if isinstance(v, tuple) else v


def fit_zmws(zmws, passes, length, world, min_batches=2):
    """largest batch (halving from `zmws`) of which `min_batches` page-locked copies (bases + pw + ipd + results) fit the host budget"""
    per_zmw = 4.6 * _mean(passes) * _mean(length) * 1.02
    budget = host_memory_budget()
    while zmws > 256 and per_zmw * zmws * min_batches * world > budget:
        zmws //= 2
    return zmws


class Job:
    """one workload through the asynchronous boundary on this rank's GPU"""

    def __init__(self, api, np, rank, world, local_rank, zmws, passes, length, distinct, steps, warmup, depth, opts, keep_sample=0):
        self.api, self.np, self.zmws, self.depth = api, np, zmws, depth
        t0 = time.time()
        first = api.synth(zmws, passes, length, seed=0xC0FFEE, first_zmw_id=rank * zmws)
        batch_bytes = 4 * first.bases.nbytes                 # bases + pw + ipd page-locked, plus slack for the result buffers
        self.nb = max(1, min(distinct, steps + warmup, host_memory_budget() // max(1, batch_bytes * world)))
        self.alg_bytes = first.algorithmic_bytes()           # 3*sum(len) + 48 + 2*L_out per ZMW
        self.sample0 = first.slice(0, min(zmws, keep_sample)) if keep_sample else None   # CPU baseline sample (pageable copy)
        self.batches = [first.pinned()]
        del first
        for i in range(1, self.nb):
            b = api.synth(zmws, passes, length, seed=0xC0FFEE, first_zmw_id=(i * world + rank) * zmws)
            self.batches.append(b.pinned())
            del b
        self.gen_s = time.time() - t0
        self.h = api.Handle(local_rank, opts=opts)
        kin = bool(opts.hifi_kinetics)

        def layout_cap(b):
            cb = b.c_struct()
            return int(api.lib().ccsx_result_layout(api.C.byref(cb), api._ptr(np.zeros(b.n_zmw + 1, np.int64), api.C.c_int64)))
        big = max(self.batches, key=layout_cap)
        # the optional float QVs (raw_qv) are not requested in the pipelined job: the HiFi record needs seq + qual + rq/ec/np/status
        self.results = [api.Results.allocate(big, kinetics=kin, pinned=True, raw=False) for _ in range(depth)]

    def run(self, nsteps, collect, t_origin=None):
        """nsteps batches through the asynchronous boundary, `depth` in flight; returns (elapsed, per-ticket timings, stats).  t_origin: the common start
        of every worker's timed region (perf_counter of the launching thread): a worker that starts late is charged for it"""
        h, depth, nb = self.h, self.depth, self.nb
        tick, kt, ok, rqsum, rqn, checks = [], [], 0, 0.0, 0, 0
        t_start = time.perf_counter() if t_origin is None else t_origin
        for k in range(nsteps + depth):
            if k >= depth:                                   # retire the oldest batch before its slot / result buffer is reused
                t_old = tick[k - depth]
                r = h.wait(t_old)
                n = self.batches[(k - depth) % nb].n_zmw
                good = r.status[:n] == 0
                ok += int(good.sum()); rqsum += float(r.rq[:n][good].sum()); rqn += int(good.sum())
                checks += int(r.seq_len[:n].sum())
                if collect:
                    kt.append(h.ticket_timings(t_old))
                h.release(t_old)
            if k < nsteps:
                b = self.batches[k % nb]
                res = self.results[k % depth]
                if b.n_zmw != len(res.status):
                    raise RuntimeError("batches must have equal ZMW counts")
                tick.append(h.submit(b, res))
        return time.perf_counter() - t_start, kt, (ok, rqsum, rqn, checks)

    def close(self):
        self.h.close()
        self.batches, self.results = [], []


STAGES = ("setup_ms", "draft_ms", "align_ms", "queue_ms", "polish_ms", "stitch_ms", "total_ms")


def stage_means(np, kt):
    return {k: float(np.mean([getattr(t, k) for t in kt])) for k in STAGES}


def kernels_span_ms(kt):
    """device time from the first kernel of the first timed batch to the last kernel of the last one (HIP events)"""
    return max(t.end_ms for t in kt) - min(t.start_ms for t in kt)


def reference_concordance(api, np, ccs_bin, sample, cores, seconds, seed=0xC0FFEE):
    """the reference tool, when the box has it (bioconda pbccs), is RUN on the same synthetic subreads — written as
    a PacBio subreads.bam by this repo's driver — and timed; its HiFi reads are compared with this library's reads.  It is the one code
    path that can pin parity; tests/test_gpu_parity.py::test_reference_concordance_harness exercises it with a stand-in `ccs` (a copy of
    this repo's driver under another path), and every way the reference can fail here — no --version, a chemistry triple it does not
    know, a crash, a time-out — ends in an `error` / `rc` field, never in an exception."""
    out = {"found": True, "path": ccs_bin}
    try:
        import bam_util
        ours = os.path.join(ROOT, "ccs_amd", "bin", "ccs")
        n = sample.n_zmw
        with tempfile.TemporaryDirectory() as td:
            sub, ref_out, our_out = os.path.join(td, "s.subreads.bam"), os.path.join(td, "ref.bam"), os.path.join(td, "ours.bam")
            passes = int(np.diff(sample.read_off)[0]); length = int(np.diff(sample.tpl_off)[0])
            subprocess.run([ours, "--write-synthetic", f"{n},{passes},{length},{seed}", sub], check=True, timeout=600)
            try:                                             # informational only: a reference without --version is still a reference
                v = subprocess.run([ccs_bin, "--version"], capture_output=True, text=True, timeout=60)
                out["version"] = (v.stdout.strip() or v.stderr.strip())[:80] if v.returncode == 0 else f"(--version: rc {v.returncode})"
            except Exception as e:
                out["version"] = f"(--version failed: {type(e).__name__})"
            t0 = time.perf_counter()
            try:
                r = subprocess.run([ccs_bin, sub, ref_out, "-j", str(cores), "--min-rq", "0.99"], capture_output=True, text=True, timeout=max(600, 40 * seconds))
            except subprocess.TimeoutExpired:
                out.update({"rc": None, "error": "reference ccs timed out"})
                return out
            dt = time.perf_counter() - t0
            out.update({"rc": r.returncode, "wall_s": round(dt, 2), "zmws": n, "zmws_per_s": round(n / dt, 3), "cores": cores,
                        "stderr_tail": r.stderr[-300:]})
            if r.returncode != 0:
                # typical: "Unsupported chemistries found" — the synthetic BAM's (BindingKit, SequencingKit, BasecallerVersion) triple is the
                # recalled Sequel II one, a real ccs may not carry it; reported, the timing above is then meaningless
                out["error"] = "reference ccs failed on the synthetic subreads (see stderr_tail" + ("; chemistry triple not supported" if "hemistr" in r.stderr else "") + ")"
                out.pop("zmws_per_s", None)
                return out
            subprocess.run([ours, sub, our_out], check=True, timeout=600)
            ref_reads = {x["tags"]["zm"]: x["seq"] for x in bam_util.read_bam(ref_out)[1]}
            our_reads = {x["tags"]["zm"]: x["seq"] for x in bam_util.read_bam(our_out)[1]}
            both = sorted(set(ref_reads) & set(our_reads))
            same = sum(1 for z in both if len(ref_reads[z]) == len(our_reads[z]) and np.array_equal(ref_reads[z], our_reads[z]))
            out.update({"hifi_reads_reference": len(ref_reads), "hifi_reads_ours": len(our_reads), "zmws_in_both": len(both),
                        "identical_sequences": same})
    except Exception as e:                                   # the concordance leg must never take the benchmark down
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node.  Plain `python bench.py --gpus N`: ONE process, N worker threads, one engine handle per "
                    "device and no collective anywhere (north_star: independent streams, no RCCL; ctypes releases the GIL inside the library) — the same "
                    "arrangement as `ccs --gpus all`.  Under torch.distributed.run (WORLD_SIZE set): one rank per GPU, WORLD_SIZE must equal N.  "
                    "CCSX_BENCH_DEVICES=0,0 (test hook) names the device of every worker, e.g. two workers on one GPU")
    ap.add_argument("--steps", type=int, default=13, help="timed steps (one batch each); the default 13 x 16384 ZMWs = twice the 100k-ZMW job of configs[1]")
    ap.add_argument("--warmup", type=int, default=3, help="untimed steps; at least one per batch slot (3), so that every hipMalloc of the engine happens before the timed region")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS), help="BASELINE.json config shape: c2 (default, the "
                    "one the metric is quoted on) 10 x 10 kb; c1 3 x 1 kb; c4 30 x 20 kb; c5 3-50 passes x 1-25 kb (log-uniform)")
    ap.add_argument("--zmws", type=int, default=0, help="ZMWs per GPU per step [workload default]")
    ap.add_argument("--passes", type=_span, default=None, help="passes per ZMW, N or LO-HI [workload default]")
    ap.add_argument("--length", type=_span, default=None, help="template length, N or LO-HI (log-uniform) [workload default]")
    ap.add_argument("--distinct", type=int, default=7, help="distinct synthetic batches the steps cycle through (bounded by host memory; 7 x 16384 ZMWs = the 100k-ZMW job of configs[1])")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight (the engine has three batch slots)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the timing barrier (nccl = RCCL; gloo for CPU-side tests)")
    ap.add_argument("--hifi-kinetics", action="store_true", help="also run the N4 kinetics kernel (not part of the headline metric)")
    ap.add_argument("--disable-heuristics", action="store_true", help="polish every position (no candidate filter): A/B for the filter's cost")
    ap.add_argument("--serial-stages", action="store_true", help="draft and polish stage on ONE compute stream (A/B for the two-stage queue)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--extra", default="c1,c4,c5", help="other BASELINE shapes reported under `extra` (N=1 only; '' = none)")
    ap.add_argument("--extra-steps", type=int, default=4)
    ap.add_argument("--pmc", action="store_true", help="counter-pass mode (under rocprofv3 --pmc): only the headline steps, no CPU baseline, no extras")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    args.passes = wl[0] if args.passes is None else args.passes
    args.length = wl[1] if args.length is None else args.length
    explicit_zmws = args.zmws > 0
    args.zmws = wl[2] if args.zmws <= 0 else args.zmws
    args.depth = max(1, min(3, args.depth))
    if args.pmc:
        args.no_cpu_baseline, args.extra = True, ""

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "WORLD_SIZE" in os.environ                    # under torch.distributed.run: one rank per GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if launched and world != args.gpus:
        # (a line that says n_gpus = N must have run on N GPUs, and the other way round)
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or run plain `python bench.py --gpus N`)")
    # worker threads of THIS process and their devices: one (the rank's device) under a launcher, N without
    if launched and world > 1:
        import torch.distributed as dist
        local_rank = int(os.environ.get("CCSX_BENCH_DEVICE", local_rank))   # test hook: several ranks on one GPU (gloo)
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        devices = [local_rank]
    else:
        env_dev = os.environ.get("CCSX_BENCH_DEVICES")
        devices = [int(x) for x in env_dev.split(",")] if env_dev else list(range(args.gpus))
        if len(devices) != args.gpus:
            sys.exit(f"bench.py: CCSX_BENCH_DEVICES names {len(devices)} devices, --gpus {args.gpus}")
        ndev = torch.cuda.device_count()
        if max(devices) >= ndev:
            sys.exit(f"bench.py: --gpus {args.gpus} needs devices {devices}, this node has {ndev} (no figure is reported for GPUs that do not exist)")
        world = args.gpus
        local_rank = devices[0]
        torch.cuda.set_device(local_rank)
    nloc = len(devices)
    if not explicit_zmws:
        args.zmws = fit_zmws(args.zmws, args.passes, args.length, world)

    import __graft_entry__ as graft
    if not os.path.exists(graft.LIB):
        graft.build()
    from ccs_amd import api

    rank_numa = None
    if launched and world > 1:
        # one rank per GPU under a launcher: the rank's process (its generator threads, page-locked batches, handle) lives on its device's NUMA node
        rank_numa = {"numa_node": int(api.lib().ccsx_device_numa_node(local_rank)), "thread_bound_to_node": int(api.lib().ccsx_bind_thread_to_device(local_rank))}

    def make_opts():
        o = api.default_opts()
        o.hifi_kinetics = 1 if args.hifi_kinetics else 0
        o.disable_heuristics = 1 if args.disable_heuristics else 0
        o.serial_stages = 1 if args.serial_stages else 0
        return o

    def barrier():
        if dist is not None:
            dist.barrier()
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    numa_bound = [None] * nloc                               # NUMA node every worker's thread was bound to (-1: not bound)

    def on_workers(fn):
        """fn(worker index) on every worker of this process at once (threads: the library calls release the GIL); results in worker order.  With several
        workers every worker thread is first bound to the CPUs of ITS device's NUMA node (ccsx_bind_thread_to_device: sysfs, no libnuma), so the page-locked
        batches it allocates and the host side of its copies are node-local.  The single worker of an N = 1 run stays unbound: the CPU
        baseline that follows uses every core."""
        if nloc == 1:
            numa_bound[0] = -1
            return [fn(0)]
        import threading
        res, err = [None] * nloc, []

        def body(i):
            try:
                numa_bound[i] = int(api.lib().ccsx_bind_thread_to_device(devices[i]))
                res[i] = fn(i)
            except BaseException as e:                       # noqa: BLE001 (re-raised below)
                err.append(e)
        th = [threading.Thread(target=body, args=(i,)) for i in range(nloc)]
        for t in th: t.start()
        for t in th: t.join()
        if err:
            raise err[0]
        return res

    # ---- the headline job: synthetic shards of every worker, distinct batches with distinct ZMW ids (deterministic), page-locked
    jobs = on_workers(lambda i: Job(api, np, rank + i, world, devices[i], args.zmws, args.passes, args.length, args.distinct, args.steps, args.warmup,
                                    args.depth, make_opts(), keep_sample=4096 if i == 0 else 0))
    job = jobs[0]
    h = job.h
    t0 = time.time()
    on_workers(lambda i: jobs[i].run(max(1, args.warmup), False))    # untimed: every hipMalloc of the engine happens here
    warm_s = time.time() - t0
    barrier()
    t_origin = time.perf_counter()                           # ONE origin for every worker's timed region
    timed = on_workers(lambda i: jobs[i].run(args.steps, True, t_origin))
    barrier()
    elapsed, kt, (ok, rqsum, rqn, checks) = timed[0]
    per_gpu = [{"worker": i, "device": devices[i], "zmws_per_s": round(args.zmws * args.steps / timed[i][0], 2), "ms_per_step": round(timed[i][0] / args.steps * 1e3, 3),
                "success_frac": timed[i][2][0] / (args.zmws * args.steps),
                "numa_node": int(api.lib().ccsx_device_numa_node(devices[i])), "thread_bound_to_node": numa_bound[i],
                "copies_hidden_frac": round(min(1.0, kernels_span_ms(timed[i][1]) * 1e-3 / timed[i][0]), 4)} for i in range(nloc)]
    if nloc > 1 and not args.pmc:
        # every worker uploads one batch AT THE SAME TIME, nothing else running: the H2D rate a GPU gets while its neighbours pull too (cross-socket staging shows here)
        def h2d(i):
            b = jobs[i].batches[0]
            t0 = time.perf_counter(); jobs[i].h.upload(b); jobs[i].h.sync()
            return (b.bases.nbytes + b.pw.nbytes + (b.ipd.nbytes if args.hifi_kinetics else 0)) / (time.perf_counter() - t0) / 1e9
        for i, g in enumerate(on_workers(h2d)):
            per_gpu[i]["h2d_GBps"] = round(g, 2)
    if nloc > 1:
        elapsed = max(t[0] for t in timed)                   # the job is done when its slowest worker is
        ok = sum(t[2][0] for t in timed) / nloc; checks = sum(t[2][3] for t in timed)
        rqsum = sum(t[2][1] for t in timed); rqn = sum(t[2][2] for t in timed)
    clocks_after = gpu_clocks(local_rank) if rank == 0 else None     # right after the last timed kernel: the clocks the run settled at
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    res0, upload_s = None, None
    if not args.pmc:
        # un-overlapped copy times of one batch (context for the pipeline: what the copies would cost in the open)
        t0 = time.time(); h.upload(job.batches[0]); h.sync(); upload_s = time.time() - t0
        h.run(); h.sync()
        res0 = h.download()                                  # also the GPU side of the CPU-baseline comparison

    if rank == 0:
        total_zmws = args.zmws * world * args.steps
        value = total_zmws / elapsed
        stage_ms = stage_means(np, kt)
        span_ms = kernels_span_ms(kt)
        names = {"draft_ms": "k_poa", "align_ms": "k_align16", "polish_ms": "k_polish", "stitch_ms": "k_stitch", "setup_ms": "k_setup"}
        dom = max(names, key=lambda k: stage_ms[k])
        alg_bytes = job.alg_bytes
        # dominant kernel: algorithmic bytes of one batch / that kernel's average launch duration in the timed region
        achieved = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9
        # HBM traffic / VALU issue / counted work of that kernel from this round's committed rocprofv3 PMC passes of the SAME command
        # (`bench.py --pmc`; (2*FETCH_SIZE + WRITE_SIZE)*1024 with the calibrated FETCH_SIZE = bytes/2, per ZMW); the file names
        # the commit it was measured at
        traffic, valu, traffic_head, traffic_zmws, traffic_src = None, None, None, None, None
        head = git_head()
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
            kz = tj["kernels"][names[dom]]
            traffic_head, traffic_zmws, traffic_src = tj.get("head"), tj.get("zmws"), tj.get("csrc_sha16")
            if args.passes == 10 and args.length == 10000 and not args.hifi_kinetics and not args.disable_heuristics:
                traffic = int(kz["hbm_bytes_per_zmw"] * args.zmws)
                valu = {k: kz[k] for k in ("valu_wave_instr_per_zmw", "valu_issue_frac_if_2cyc", "valu_issue_frac_if_4cyc", "lanes_active_frac",
                                           "valu_cycles_from_isa_histogram", "valu_issue_frac_from_isa_histogram", "cell_updates_per_zmw") if k in kz}
                # (no figure fitted to the counters it is compared with — the two bounds, and the kernel's own opcode histogram x the single-opcode table)
                valu["calibration"] = "profiles/r06_valu_peak.txt: VOP2 / three-source FMA float ops and v_add_u32 issue in ~2.5 SIMD cycles per wave64, integer max, compares, selects and DPP ops in ~4.3"
        except Exception:
            traffic, valu = None, None
        roofline = {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_over_algorithmic": round(traffic / alg_bytes, 2) if traffic else None,
                    "traffic_source": f"profiles/{TRAFFIC_FILE} (rocprofv3 --pmc passes of `bench.py --pmc --zmws {traffic_zmws}`, measured at commit {traffic_head})" if traffic else None,
                    # a counter file from another commit is flagged, not trusted blindly: same commit, or — where git can
                    # tell — no change under ccs_amd/csrc since
                    "traffic_head": traffic_head, "traffic_is_this_build": (bool(head) and bool(traffic_head) and head.split("+")[0] == str(traffic_head).split("+")[0]) if traffic else None,
                    "kernels_changed_since_traffic": kernels_changed_since(traffic_head) if traffic else None,
                    "traffic_matches_these_sources": (traffic_src == csrc_sha16()) if (traffic and traffic_src) else None,   # sha256 over ccs_amd/csrc + include/ccsx.h
                    "avg_launch_ms": round(stage_ms[dom], 3), "algorithmic_bytes_per_launch": alg_bytes, "valu": valu,
                    "note": "DP matrices stay in LDS/registers; arithmetic intensity ~kFLOP/B so the HBM fraction is <<1% by construction (SURVEY.md 8d); "
                            "the kernel is bound by VALU issue and dependent-chain latency (DESIGN.md 4)"}
        c2 = args.workload == "c2" and args.passes == 10 and args.length == 10000
        switches = api.lib().ccsx_runtime_switches().decode()     # CCSX_* scheduling overrides the library found in the environment ("" = none)
        env_serial = os.environ.get("CCSX_SERIAL_STAGES")
        serial_effective = (int(env_serial) != 0) if (env_serial is not None and env_serial.lstrip("-").isdigit()) else bool(args.serial_stages)
        if switches:
            print(f"bench.py: runtime switches in effect: {switches}", file=sys.stderr)
        out = {
            "metric": "ZMWs/sec (HiFi reads/sec), 10-pass x 10 kb synthetic" if c2 else f"ZMWs/sec, {args.passes} passes x {args.length} bp synthetic",
            "value": round(value, 2), "unit": "ZMWs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            **({"per_gpu": per_gpu, "multi_gpu": "one process, one worker thread and one engine handle per device, no collective (value = all workers' ZMWs / the slowest worker's time)"} if nloc > 1 else {}),
            **({"rank0_numa": rank_numa, "multi_gpu": "one rank per GPU (torch.distributed.run), every rank bound to its device's NUMA node, no collective on the data path (barrier + max over ranks for the timing only)"} if rank_numa else {}),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.passes} passes x {args.length} bp synthetic subreads (BASELINE configs[{int(args.workload[1]) - 1}] shape), "
                                   f"{args.zmws} ZMWs per GPU per step, {job.nb} distinct batches, host-pinned -> H2D -> kernels -> D2H, {args.depth} batches in flight",
                       "preset": args.workload, "zmws_per_gpu": args.zmws, "distinct_batches": job.nb, "in_flight": args.depth, "passes": args.passes,
                       "template_len": args.length, "parallelism": f"zmw-shard x{world}" + ("" if launched or world == 1 else " (worker threads of one process)"), "model": "SYN-1",
                       "hifi_kinetics": bool(args.hifi_kinetics), "candidate_filter": not args.disable_heuristics,
                       # what the LIBRARY did, not what this script asked for: the environment can override the option
                       "stages": "serial (one compute stream)" if serial_effective else "draft stage of batch k+1 under the polish stage of batch k (two compute streams)",
                       "runtime_switches": switches, "spec_version": int(api.lib().ccsx_spec_version())},
            "timed_region": "first ccsx_submit to last ccsx_wait: pinned host -> H2D -> kernels -> D2H (PCIe-inclusive); downloaded per ZMW: status, sequence, phred QVs, rq, ec, np, fn/rn, iterations (the optional float QVs are not requested)",
            "resident_zmws_per_s": round(args.zmws * args.steps / (span_ms * 1e-3), 2),
            "kernels_ms_per_step": round(span_ms / args.steps, 3),
            "roofline": roofline,
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "stage_ms_note": "per batch, HIP events on the stream of each stage; with two compute streams the stages of consecutive batches overlap, "
                             "so ms_per_step < draft + align + polish + stitch; queue_ms = the batch waiting between the stages",
            "success_frac": ok / (args.zmws * args.steps), "mean_rq": rqsum / rqn if rqn else None, "consensus_bases": checks,
            "host": {"synth_s": round(job.gen_s, 2), "warmup_s": round(warm_s, 2), "unoverlapped_upload_s": round(upload_s, 3) if upload_s else None,
                     "copies_hidden_frac": round(min(1.0, span_ms * 1e-3 / elapsed), 4),
                     "bam_pipeline_note": "this bench feeds the engine from memory; the `ccs` driver's BAM side costs 0.85-0.94 CPU-s per 1000 ZMWs "
                                          "(BGZF inflate 0.57), i.e. ~28 host cores per MI355X at the engine's rate: 17.0-18.8k ZMWs/s BAM->BAM and 24-28k for "
                                          "the host side alone on this box's 16 usable threads (profiles/r03_cli_host_pipeline.txt, DESIGN.md 7)"},
            "head": head,
            "gpu_clocks": {"after_timed_region": clocks_after, "source": "rocm-smi --showclocks --showpower --showmaxpower --showperflevel"},
        }
        cores = effective_cores()
        # the reference tool, if the box has it (expected absent; bioconda pbccs): probed, and run when found
        ccs_bin = shutil.which("ccs")
        if ccs_bin and os.path.realpath(ccs_bin) == os.path.realpath(os.path.join(ROOT, "ccs_amd", "bin", "ccs")):
            ccs_bin = None                                   # this repo's own driver is not the reference
        if ccs_bin and not args.pmc and c2:
            out["reference_ccs"] = reference_concordance(api, np, ccs_bin, job.sample0.slice(0, min(job.sample0.n_zmw, 4 * cores)), cores, args.cpu_seconds)
        else:
            out["reference_ccs"] = {"found": bool(ccs_bin), "path": ccs_bin, "note": "command -v ccs on this box: absent, nothing to run" if not ccs_bin else "present; concordance runs with the default c2 workload"}
        if not args.no_cpu_baseline and world == 1:          # reported baseline: rank 0 at N=1 only
            import oracle_lib
            sample0 = job.sample0
            probe = sample0.slice(0, 1)
            pr = api.Results.allocate(probe)
            t1 = time.perf_counter()
            oracle_lib.consensus_batch(h.model, h.opts, probe, pr, nthreads=1)
            t1 = time.perf_counter() - t1
            n_s = int(min(sample0.n_zmw, max(cores, round(args.cpu_seconds * cores / max(t1, 1e-3)))))
            n_s = min(sample0.n_zmw, max(cores, (n_s // cores) * cores))    # whole rounds of one ZMW per thread
            sample = sample0.slice(0, n_s)
            sr = api.Results.allocate(sample)
            oracle_lib.counts_reset()
            t2 = time.perf_counter()
            oracle_lib.consensus_batch(h.model, h.opts, sample, sr, nthreads=cores)
            t2 = time.perf_counter() - t2
            cnt = oracle_lib.counts()
            same = all(np.array_equal(sr.sequence(z), res0.sequence(z)) for z in range(n_s))
            qv_max = max((float(np.max(np.abs(sr.raw(z) - res0.raw(z)))) if len(sr.raw(z)) else 0.0) for z in range(n_s)) if same else None
            core_s = t2 * cores / n_s
            out["cpu_baseline"] = {"value": round(n_s / t2, 3), "unit": "ZMWs/s", "cores": cores, "kind": "port",
                                   "sample": f"first {n_s} ZMWs of batch 0, oracle/ccs_oracle.c with OpenMP over ZMWs "
                                             f"({t2:.1f} s wall; single-thread probe {t1:.2f} s/ZMW); reference ccs binary "
                                             f"{'found at ' + ccs_bin if ccs_bin else 'not on PATH (probed)'}",
                                   "gpu_matches_cpu_sequences": bool(same), "max_abs_qv_diff": qv_max,
                                   "core_seconds_per_zmw": round(core_s, 4),
                                   "gpu_equivalent_cores": round(value * core_s, 1),     # host cores of THIS port one MI355X replaces (value x core-s per ZMW)
                                   "context": f"this SPEC's port costs {core_s:.3f} core-s per ZMW here; docs/img/runtime.png shows ~1 core-s for ccs 4.2 at "
                                              "10 kb x 7 passes, i.e. the port does far less CPU work per ZMW than ccs, so the GPU/CPU ratio is not a "
                                              "statement about ccs (PacBio's own GPU claim: 10x over 128 cores, docs/faq/revio.md:23-25)"}
            out["speedup_vs_cpu_all_cores"] = round(value / (n_s / t2), 2)
            # SURVEY.md §8d secondary figure: COUNTED DP cell updates per ZMW (the oracle's instrumented path on the same ZMWs), the rate the
            # GPU sustains over them, and the VALU lane-operations the kernels spend per cell (PMC passes, when the traffic file has them)
            cells = {k: cnt[k] / max(1, cnt["zmws"]) for k in ("cells_poa", "cells_align", "cells_fill", "cells_score")}
            tot = sum(cells.values())
            work = {"cell_updates_per_zmw": {k[6:]: int(v) for k, v in cells.items()}, "cell_updates_per_zmw_total": int(tot),
                    "gpu_cell_updates_per_s": round(tot * value, 1),
                    "note": "cells of the banded DP columns (POA: 32 rows x in-edges, alignment: 16 / 64 rows), of the alpha + beta matrices of every "
                            "(read, window, round), and of the banded mutation links; counted, not estimated"}
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
                vt = sum(kz.get("valu_wave_instr_per_zmw", 0) for kz in tj["kernels"].values())
                if vt:
                    work["valu_lane_ops_per_cell"] = round(vt * 64 / tot, 1)
            except Exception:
                pass
            out["roofline"]["work"] = work
            # the interpretable roofline: counted cell updates x the nominal 8 operations per cell against the non-packed
            # VALU issue peak of the chip — what share of the machine's lane-operations is the recurrence itself
            out["roofline"]["valu_algorithmic_frac"] = round(tot * NOMINAL_OPS_PER_CELL * value / VALU_PEAK_LANE_OPS, 4)
            out["roofline"]["valu_algorithmic_note"] = (f"{int(tot)} counted cell updates per ZMW x {NOMINAL_OPS_PER_CELL} nominal ops x {value:.0f} ZMWs/s / "
                                                        f"{VALU_PEAK_LANE_OPS:.3g} lane-ops/s (1024 SIMDs x 32 lanes/cycle x 2.4 GHz)")
        for j in jobs: j.close()
        # ---- the other BASELINE shapes through the same pipeline (N=1 only; a few steps each)
        if world == 1 and args.extra:
            extra = {}
            for name in [x for x in args.extra.split(",") if x and x != args.workload]:
                try:
                    p_, l_, z_ = WORKLOADS[name]
                    z_ = fit_zmws(z_, p_, l_, 1)
                    j2 = Job(api, np, 0, 1, local_rank, z_, p_, l_, 2, args.extra_steps, args.depth, args.depth, make_opts())
                    j2.run(args.depth, False)            # one untimed step per batch slot: all device buffers exist before the timed steps
                    torch.cuda.synchronize()
                    el, kt2, (ok2, rs2, rn2, _) = j2.run(args.extra_steps, True)
                    sm = stage_means(np, kt2)
                    dom2 = max(names, key=lambda k: sm[k])
                    ach2 = j2.alg_bytes / (sm[dom2] * 1e-3) / 1e9
                    traffic2 = None
                    try:
                        t2j = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILES_EXTRA[name])))
                        traffic2 = {"bytes_per_launch": int(t2j["kernels"][names[dom2]]["hbm_bytes_per_zmw"] * z_), "whole_step_bytes_per_zmw": int(sum(k.get("hbm_bytes_per_zmw", 0) for k in t2j["kernels"].values())),
                                    "source": f"profiles/{TRAFFIC_FILES_EXTRA[name]}", "head": t2j.get("head"), "matches_these_sources": t2j.get("csrc_sha16") == csrc_sha16()}
                    except Exception:
                        pass
                    extra[name] = {"workload": f"{p_} passes x {l_} bp", "zmws_per_step": z_, "steps": args.extra_steps,
                                   **({"metric": "ZMWs/s (0 HiFi reads: three passes never reach --min-rq 0.99; BASELINE configs[0] is a plumbing shape)"} if name == "c1" else {}),
                                   "value": round(z_ * args.extra_steps / el, 2), "unit": "ZMWs/s",
                                   "resident_zmws_per_s": round(z_ * args.extra_steps / (kernels_span_ms(kt2) * 1e-3), 2),
                                   "stage_ms": {k: round(v, 3) for k, v in sm.items()},
                                   "success_frac": round(ok2 / (z_ * args.extra_steps), 4), "mean_rq": rs2 / rn2 if rn2 else None,
                                   "algorithmic_bytes_per_launch": j2.alg_bytes,
                                   # the same roofline object as the headline's, for this shape's dominant kernel
                                   "roofline": {"bound": "hbm", "kernel": names[dom2], "achieved": round(ach2, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(ach2 / 8000.0, 6),
                                                "avg_launch_ms": round(sm[dom2], 3), "traffic": traffic2["bytes_per_launch"] if traffic2 else None, "traffic_detail": traffic2}}
                    if name == "c1" and not args.no_cpu_baseline:
                        # BASELINE configs[0]: "1k synthetic ZMWs, 3 passes x 1 kb, reference `ccs --num-threads=1` on CPU" — the reference is absent (docs-only
                        # mount), so the figure SURVEY.md 8d asks for beside the GPU line is the port on ONE thread over a bounded sample of the same ZMWs
                        import oracle_lib
                        s1 = api.synth(min(1000, z_), p_, l_, seed=0xC0FFEE, first_zmw_id=0)
                        r1 = api.Results.allocate(s1)
                        tc = time.perf_counter()
                        oracle_lib.consensus_batch(j2.h.model, j2.h.opts, s1, r1, nthreads=1)
                        tc = time.perf_counter() - tc
                        extra[name]["cpu_1thread"] = {"value": round(s1.n_zmw / tc, 2), "unit": "ZMWs/s", "cores": 1, "kind": "port",
                                                      "sample": f"the first {s1.n_zmw} ZMWs of the c1 job (BASELINE configs[0] is 1k ZMWs), oracle/ccs_oracle.c on one thread, {tc:.2f} s"}
                    j2.close()
                    del j2
                except Exception as e:
                    extra[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            try:                                          # predicted vs empirical accuracy (tools/qv_calibration.py; CPU restatement, committed)
                qc = json.load(open(os.path.join(ROOT, "profiles", QVCAL_FILE)))
                extra["qv_calibration"] = {"source": f"profiles/{QVCAL_FILE}", "spec_version": qc.get("spec_version"), "datasets": qc.get("headline")}
            except Exception:
                pass
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    else:
        for j in jobs: j.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
