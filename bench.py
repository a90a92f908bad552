#!/usr/bin/env python3
"""bench.py — llama-bench's pp512 / tg128 protocol over the offloaded surface (quantized MUL_MAT / MUL_MAT_ID).

One *step* is one llama-bench repetition (examples/llama-bench/llama-bench.cpp:1430-1468, 1620-1642) restricted
to the hot path: a 512-token prompt batch through every MUL_MAT of the model (pp512), then 128 single-token
passes (tg128).  Weights are synthetic (random valid quant blocks of the model's shapes and types), activations
synthetic, everything resident in HBM before the timed region.  tok/s = tokens / time of that part, as llama-bench
computes it; `value` is the tg128 rate (the HBM-bound mat-vec path), `pp512_tok_s` is reported beside it.

    python bench.py [--gpus N --steps K --warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 = ggml row split: every weight's rows are sharded over the ranks (one process per GPU) and the partial
results are concatenated with RCCL all-gather over xGMI after each MUL_MAT group ("scaling": "strong").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16/f16 MFMA peak ~2.5 PF (spec)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def launch_class(grp):
    """which launch of a layer a group is: the classes VERDICT r2 asked to see separately"""
    n = grp.mats[0].name.split(".")[-1]
    return {"attn_q": "qkv", "attn_output": "wo", "ffn_gate": "gate_up", "ffn_down": "down", "ffn_gate_exps": "gate_up_exps",
            "ffn_down_exps": "down_exps", "output": "output"}.get(n, n)


def roofline_leg(hp, q, n_batch, torch, n_outputs=None):
    """HIP-event timing of the pass's launches.  Every group is issued through HotPath.group_call, i.e. by the very closure the timed
    pass runs, and is bucketed by (launch class, the kernels that call issued): the labels come from the library's own launch
    sites (qmm_trace_begin / qmm_trace_end), not from a replica of its dispatch rules.  The model's 32 layers give every bucket
    16-32 distinct weight sets, so a bucket is issued back to back on rotating weights (no Infinity-Cache reuse) inside ONE event
    pair on the launch stream, behind a spin kernel that keeps the host ahead of the GPU.  Bucket time / launches = average
    launch duration including the kernel-to-kernel boundary (not a per-launch event cost).
    Returns {(class, labels): [algorithmic bytes, flops, seconds, calls]}."""
    buckets = {}
    for grp in hp.wl.groups:
        n_tokens = hp.wl.group_tokens(grp, n_batch, n_outputs)          # the groups computed on the output rows only run at n_outputs
        fn = hp.group_call(grp, hp.prepare(n_tokens))
        labels = q.trace(fn)                                            # (also the warm-up call of this weight set)
        m0 = grp.mats[0]
        nbytes = sum(m.algo_bytes(n_tokens) for m in grp.mats) - (len(grp.mats) - 1) * n_tokens * m0.K * 4 * (1 if not m0.n_expert else 0)
        if m0.n_expert and len(grp.mats) == 2:
            nbytes -= n_tokens * m0.K * 4                               # the twin MUL_MAT_IDs share src1 as well
        if hp.world > 1:                                                # this rank's shard: weights and dst columns are split, src1 is not
            nbytes = n_tokens * m0.K * 4 + sum(hp.weights[m.name][0].numel() + n_tokens * hp.weights[m.name][0].shape[0] * 4 for m in grp.mats)
        fl = sum(m.flops(n_tokens) for m in grp.mats) // hp.world
        buckets.setdefault((launch_class(grp), labels), []).append((fn, nbytes, fl))
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2.0e8))
    evs = {}
    for key, items in buckets.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for fn, _, _ in items:
            fn()
        e1.record()
        evs[key] = (e0, e1)
    torch.cuda.synchronize()
    agg = {}
    for key, items in buckets.items():
        e0, e1 = evs[key]
        agg[key] = [sum(i[1] for i in items), sum(i[2] for i in items), e0.elapsed_time(e1) * 1e-3, len(items)]
    return agg


def by_kernel(agg):
    """token generation: one group is one launch, so a bucket's label IS its kernel; merge the classes that ran the same one"""
    out = {}
    for (cls, labels), (nb, fl, sec, cnt) in agg.items():
        k = " + ".join(labels)
        a = out.setdefault(k, [0, 0, 0.0, 0, 0, []])
        a[0] += nb; a[1] += fl; a[2] += sec; a[3] += cnt; a[4] += cnt * len(labels); a[5].append(cls)
    return out


class LlamaBench:
    """The reference's own libllama (oracle/_ref, compiled unmodified from /root/reference) driven through llama-bench's protocol
    (examples/llama-bench/llama-bench.cpp:1430-1468, 1605-1642; tests/cpp/llama_e2e.cpp): warm-up, llama_kv_self_clear, one llama_decode
    of 512 random tokens, 128 single-token decodes with a synchronize each.  Always a child process (never an exec of this one, which
    holds the GPU).  `gpu` loads the MI355X module through GGML_BACKEND_PATH with every layer offloaded; otherwise -ngl 0 on the ggml
    CPU backend, which is the CPU baseline SURVEY 8(d) defines.  The synthetic GGUF is written once and shared by both legs."""

    def __init__(self, name):
        import shutil
        import tempfile
        self.name = name
        self.exe = ROOT / "oracle" / "_ref" / "llama-e2e"
        self.plugin = ROOT / "ggml-hexagon_amd" / "libggml-mi355x.so"
        self.ok = self.exe.exists() and not name.startswith("llama3-70b") and shutil.disk_usage(tempfile.gettempdir()).free >= (60 << 30 if name.startswith("mixtral") else 12 << 30)
        self.tmp = tempfile.mkdtemp(prefix="qmm_bench_") if self.ok else None
        self.gguf = None

    def _run(self, args, env, timeout):
        import subprocess
        p = subprocess.run([str(self.exe), *args], check=True, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(self.exe.parent))
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        return (json.loads(lines[-1]) if lines else None), p.stderr

    def _env(self, gpu, extra=None):
        env = {k: v for k, v in os.environ.items() if k != "GGML_BACKEND_PATH"}
        if gpu:
            env["GGML_BACKEND_PATH"] = str(self.plugin)
        env.update(extra or {})
        return env

    def write(self):
        if self.gguf is None:
            self.gguf = os.path.join(self.tmp, "m.gguf")
            self._run(["write", "--config", self.name, "--gguf", self.gguf], self._env(False), 300)
        return self.gguf

    def run(self, gpu, threads, n_prompt=512, n_gen=128, reps=3, timeout=300):
        if not self.ok or (gpu and not self.plugin.exists()):
            return None
        try:
            gguf = self.write()
            r, _ = self._run(["bench", "--gguf", gguf, "--ngl", "99" if gpu else "0", "-p", str(n_prompt), "-n", str(n_gen), "-r", str(reps), "-t", str(threads)],
                             self._env(gpu), timeout)
            out = {"pp512_tok_s": r["pp_tok_s"], "tg128_tok_s" if n_gen == 128 else "tg_tok_s": r["tg_tok_s"], "reps": reps, "threads": threads,
                   "n_gen": n_gen, "devices": r.get("devices")}
            if gpu:
                # the same protocol once more with the module timing every graph on its stream (event pair around the launches of a
                # graph_compute): GPU time per token / per prompt batch, i.e. what of the end-to-end time is not the host's
                _, err = self._run(["bench", "--gguf", gguf, "--ngl", "99", "-p", str(n_prompt), "-n", "64", "-r", "1", "-t", str(threads)],
                                   self._env(True, {"GGML_MI355X_TIMING": "1"}), timeout)
                import re
                m = re.search(r"tg graphs (\d+) stream_ms ([0-9.]+) \| pp graphs (\d+) tokens (\d+) stream_ms ([0-9.]+)(?: min_ms ([0-9.]+))?", err)
                if m:
                    ntg, mtg, npp, tpp, mpp = int(m.group(1)), float(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))
                    out["gpu_ms_per_token"] = round(mtg / max(ntg, 1), 4)
                    # the fastest prompt graph: the warm-up pass also re-lays weights at their first use (and loads code objects)
                    out["gpu_ms_per_prompt_batch"] = round(float(m.group(6)) if m.group(6) else mpp / max(npp, 1), 3)
                h = re.search(r"host us per tg graph: outside graph_compute ([0-9.]+) \| reader analysis ([0-9.]+) \| issue loop ([0-9.]+) \| waiting in synchronize ([0-9.]+)", err)
                if h:       # where the host's wall time goes around a one-token graph (the GPU idles during `outside` + `analysis`)
                    out["host_us_per_token"] = {"outside_graph_compute": float(h.group(1)), "reader_analysis": float(h.group(2)),
                                                "issue_loop": float(h.group(3)), "waiting_in_synchronize": float(h.group(4))}
            return out
        except Exception as e:              # reported-only
            return {"pp512_tok_s": None, "tg128_tok_s": None, "threads": threads, "error": f"{type(e).__name__}: {str(e)[-300:]}"}

    def close(self):
        import shutil
        if self.tmp:
            shutil.rmtree(self.tmp, ignore_errors=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, cores, lb):
    """the same llama-bench protocol with -ngl 0 (the reference's libllama on its ggml CPU backend, threads = this
    box's CPU share) is the baseline `value`.  Beside it, under `matmul_only`, the MUL_MAT path alone: the reference's CPU backend
    (oracle/_ref) or, failing that, our C port, timed on a bounded sample: each distinct (type, K, M) once at N=1 and N=512,
    summed over the model"""
    import numpy as np
    from oracle.pyoracle import Oracle, RefGgml, ref_available
    from ggml_hexagon_amd import synth
    kind = "reference" if ref_available() else "port"
    ref = RefGgml() if kind == "reference" else None
    orc = None if ref else Oracle()
    mats = [m for m in wl.all_mats() if not m.n_expert]
    per_layer = (len(mats) - 1) // wl.n_layer
    shapes, shapes_out = {}, {}           # (type, K, M) -> count; of which in groups a prompt batch runs at n_outputs = 1
    for g in wl.groups:
        for m in g.mats:
            if m.n_expert:
                continue
            shapes[(m.type, m.K, m.M)] = shapes.get((m.type, m.K, m.M), 0) + 1
            if g.outputs_only:
                shapes_out[(m.type, m.K, m.M)] = shapes_out.get((m.type, m.K, m.M), 0) + 1
    t_tg, t_pp, spent = 0.0, 0.0, time.perf_counter()
    rng = np.random.default_rng(0)
    for (t, K, M), count in shapes.items():
        Ms = min(M, 16384)                    # the 128k-row output matrix is sampled by rows
        w = synth.synth_weights(t, Ms, K, seed=1)
        x1 = rng.uniform(-1, 1, (1, K)).astype(np.float32)
        if ref:
            _, dt = ref.graph_mul_mat(t, w, K, x1, n_threads=cores, repeat=3)
        else:
            t0 = time.perf_counter(); orc.mul_mat(t, w, K, x1); dt = time.perf_counter() - t0
        t_tg += dt * (M / Ms) * count
        n_out = shapes_out.get((t, K, M), 0)                  # these run at N = 1 in the prompt pass too
        t_pp += dt * (M / Ms) * n_out
        count -= n_out
        if count == 0:
            continue
        if time.perf_counter() - spent < 25.0:
            xp = rng.uniform(-1, 1, (512, K)).astype(np.float32)
            Mp = min(Ms, 4096)
            if ref:
                _, dp = ref.graph_mul_mat(t, w[:Mp], K, xp, n_threads=cores, repeat=1)
            else:
                t0 = time.perf_counter(); orc.mul_mat(t, w[:Mp], K, xp[:32]); dp = (time.perf_counter() - t0) * 16
            t_pp += dp * (M / Mp) * count
        else:
            t_pp = float("nan")
    threads = min(cores, 16)
    e2e = lb.run(False, threads, n_gen=16, reps=1) if lb is not None else None
    mm = {"tg128_tok_s": round(1.0 / t_tg, 3), "pp512_tok_s": None if t_pp != t_pp else round(512.0 / t_pp, 2), "threads": cores,
          "kind": kind, "variant": getattr(ref, "variant", "scalar+omp"),
          "sample": f"each distinct (type,K,M) of {wl.name} once at N=1 (x3) and N=512 (rows capped at 16384/4096, scaled), "
                      f"summed over the model's {len(mats)} MUL_MATs ({per_layer} per layer); prompt pass as llama-bench runs it: "
                      f"last layer's FFN and the output projection at n_outputs = 1"}
    if e2e and e2e.get("tg_tok_s"):
        return {"value": e2e["tg_tok_s"], "unit": "tok/s (tg, llama-bench protocol, -ngl 0)", "pp512_tok_s": e2e["pp512_tok_s"], "cores": threads,
                "kind": "reference", "cpu": cpu_model(),
                "sample": "the reference's libllama on the ggml CPU backend, whole model end to end: warm-up, one pp512, 16 generated tokens (1 rep)",
                "matmul_only": mm}
    return {"value": mm["tg128_tok_s"], "unit": "tok/s (tg128, MUL_MAT path only)", "pp512_tok_s": mm["pp512_tok_s"], "cores": cores, "kind": kind,
            "cpu": cpu_model(), "sample": mm["sample"], "llama_bench_protocol": e2e}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("QMM_WORKLOAD", "llama3-8b-q4_k_m"))
    ap.add_argument("--n-prompt", type=int, default=512)
    ap.add_argument("--n-gen", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs (llama-bench protocol through libllama: device and CPU)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--wire-layout", action="store_true", help="keep Q4_0 / Q8_0 / Q6_K weights in GGUF wire layout (no planar repack, SURVEY 8f-2)")
    ap.add_argument("--chain", action="store_true", help="token generation as persistent chains (csrc/qmm_chain.hiph) instead of one launch per MUL_MAT group")
    ap.add_argument("--all-logits", action="store_true",
                    help="prompt pass with logits for every token (n_outputs = n_prompt) instead of llama-bench's last-token-only")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # rehearsal aids for a one-GPU box (the driver's multi-GPU runs use neither): QMM_BENCH_DEVICE pins every rank to one
    # device, QMM_BENCH_DIST_BACKEND=gloo replaces RCCL, which refuses two ranks on one device
    if os.environ.get("QMM_BENCH_DEVICE"):
        local = int(os.environ["QMM_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_split = world == 1 and os.environ.get("QMM_BENCH_FORCE_SPLIT") == "1"    # rehearsal: a world of one through RCCL and the captured exchange
    if force_split:
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("MASTER_PORT", "29571")
    if world > 1 or force_split:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("QMM_BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from ggml_hexagon_amd import workload
    from ggml_hexagon_amd.capi import Qmm
    from ggml_hexagon_amd.hotpath import HotPath
    from ggml_hexagon_amd.rowsplit import RowConcat

    q = Qmm(local)
    wl = workload.get(args.workload)
    concat = RowConcat(always_collective=force_split) if dist is not None else None
    hp = HotPath(q, wl, dev, rank, world, concat, planar=not args.wire_layout)
    hp.chain = args.chain
    hp.prepare(args.n_prompt)
    hp.prepare(1)
    # llama-bench's prompt test wants the last token's logits only (llama_batch_get_one: batch.logits = NULL ->
    # n_outputs_all = 1, src/llama-context.cpp:1232-1244): the last layer's FFN and the output projection then run on
    # one row (ggml_get_rows(cur, inp_out_ids), src/llama-model.cpp:4270-4275)
    n_out_pp = None if args.all_logits else 1
    # token generation replays a hipGraph of the pass.  With a row split the RCCL all-gathers are captured with the launches
    # (tests/test_gpu_rccl.py rehearses that on one GPU); QMM_BENCH_GRAPH_MULTI=0 keeps N > 1 eager.  A capture that fails on every
    # rank alike falls back to eager launches instead of ending the run.
    use_graph = not args.no_graph and (world == 1 or (os.environ.get("QMM_BENCH_GRAPH_MULTI", "1") != "0" and
                                                     os.environ.get("QMM_BENCH_DIST_BACKEND", "nccl") == "nccl"))   # (gloo stages through the host: not capturable)
    graph = None
    if use_graph:
        try:
            graph = hp.capture(1)
        except Exception as e:                       # noqa: BLE001
            if world == 1:
                raise
            print(f"[bench] rank {rank}: hipGraph capture of the split pass failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
        if dist is not None:                         # all ranks take the same path
            ok = torch.tensor([1 if graph is not None else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                graph = None

    # the prompt pass as a hipGraph too (one GPU): its ~430 launches leave a Python host ~25 us each, which a slow or shared host
    # does not have (round 3: the same kernels gave 11.9 ms GPU-bound and 21.9 ms issued eagerly from one box's host)
    graph_pp = None
    if use_graph and world == 1 and not force_split and os.environ.get("QMM_BENCH_GRAPH_PP", "1") != "0":
        try:
            graph_pp = hp.capture(args.n_prompt, n_out_pp)
        except Exception as e:                       # noqa: BLE001
            print(f"[bench] hipGraph capture of the prompt pass failed ({type(e).__name__}: {e}); running it eager", file=sys.stderr)
            graph_pp = None
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(e=None):
        if e: e[0].record()
        if graph_pp is not None:
            graph_pp.replay()
        else:
            hp.run(args.n_prompt, n_out_pp)
        if e: e[1].record()
        for _ in range(args.n_gen):
            if graph is not None:
                graph.replay()
            else:
                hp.run(1)
        if e: e[2].record()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(ev[i])
    barrier()
    wall = time.perf_counter() - t0

    pp_s = sum(e[0].elapsed_time(e[1]) for e in ev) * 1e-3 / args.steps
    tg_s = sum(e[1].elapsed_time(e[2]) for e in ev) * 1e-3 / args.steps
    times = torch.tensor([wall, pp_s, tg_s], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    wall, pp_s, tg_s = times.tolist()

    tg_tok_s = args.n_gen / tg_s
    pp_tok_s = args.n_prompt / pp_s

    # ---- roofline leg: per-launch HIP events (this rank's shard of the work)
    agg1 = roofline_leg(hp, q, 1, torch)
    aggp = roofline_leg(hp, q, args.n_prompt, torch, n_out_pp)

    # token generation: per kernel (a one-token group is ONE launch; buckets of different launch classes that ran the same kernel merge)
    kern1 = by_kernel(agg1)
    k1 = max(kern1, key=lambda k: kern1[k][2])
    nb1, _, s1, calls1, launches1, classes1 = kern1[k1]
    traffic, traffic_note = None, None
    tf = ROOT / "profiles" / "pmc_traffic.json"
    if tf.exists():
        # HBM bytes per launch from the PMC passes (profiles/tools/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes over THIS
        # leg's dispatch population, gfx950 corrections applied): taken only from a file that names this kernel and this workload
        try:
            tj = json.loads(tf.read_text())
            if tj.get("label") == k1 and tj.get("workload") == wl.name:
                traffic = tj.get("bytes_per_launch")
                traffic_note = {"algo_bytes_per_launch_of_that_population": tj.get("algo_bytes_per_launch"), "ratio": tj.get("ratio"), "dispatches": tj.get("dispatches")}
        except Exception:
            traffic = None
    roof = {"bound": "hbm", "kernel": "qmm::" + k1,
            "achieved": round(nb1 / s1 / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nb1 / s1 / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_population": traffic_note, "launches": launches1, "launch_classes": sorted(set(classes1)),
            "avg_launch_us": round(s1 / launches1 * 1e6, 2), "algo_bytes_per_launch": int(nb1 / launches1),
            "all_tg_launches_GBs": round(sum(a[0] for a in agg1.values()) / sum(a[2] for a in agg1.values()) / 1e9, 1),
            "kernels": {k: {"launches": v[4], "avg_launch_us": round(v[2] / v[4] * 1e6, 2), "GBs": round(v[0] / v[2] / 1e9, 1),
                            "frac": round(v[0] / v[2] / 1e9 / HBM_PEAK_GBS, 4), "classes": sorted(set(v[5]))} for k, v in kern1.items()}}
    # prompt pass: per launch class (qkv / wo / gate_up / down), each INCLUDING its prep_act and splitk_reduce launches; the groups
    # llama-bench's prompt pass runs on one row (last layer's FFN, output projection) are mat-vec launches and listed apart
    cls_pp = {}
    for (cls, labels), (nb, fl, sec, cnt) in aggp.items():
        batched = any(l.startswith(("mfma_", "prep_act")) for l in labels)
        c = cls_pp.setdefault((cls, batched), {"flops": 0, "s": 0.0, "calls": 0, "launches": 0, "kernels": set()})
        c["flops"] += fl; c["s"] += sec; c["calls"] += cnt; c["launches"] += cnt * len(labels); c["kernels"].update(labels)
    pp_classes = {}
    for (cls, batched), c in cls_pp.items():
        if not batched:
            continue
        pp_classes[cls] = {"TFLOPs": round(c["flops"] / c["s"] / 1e12, 1), "frac": round(c["flops"] / c["s"] / 1e12 / MFMA_PEAK_TFLOPS, 4),
                           "us_per_call": round(c["s"] / c["calls"] * 1e6, 2), "calls": c["calls"], "launches_per_call": round(c["launches"] / c["calls"], 2),
                           "share_of_pp_time": None, "kernels": sorted(c["kernels"])}
    tot_pp_s = sum(c["s"] for (cls, batched), c in cls_pp.items())
    for (cls, batched), c in cls_pp.items():
        if batched:
            pp_classes[cls]["share_of_pp_time"] = round(c["s"] / tot_pp_s, 4)
    bat = [(k, c) for k, c in cls_pp.items() if k[1]]
    flp, sp = sum(c["flops"] for _, c in bat), sum(c["s"] for _, c in bat)
    (kp, _), cdom = max(bat, key=lambda kc: kc[1]["s"])
    roof_pp = {"bound": "mfma", "kernel": "all batched launches of the prompt pass, each class with its prep_act / splitk_reduce launches",
               "achieved": round(flp / sp / 1e12, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flp / sp / 1e12 / MFMA_PEAK_TFLOPS, 4),
               "dominant_class": kp, "classes": pp_classes}

    out = {
        "metric": "llama-bench pp512 & tg128 tok/s over the offloaded quantized MUL_MAT/MUL_MAT_ID path (value = tg128)",
        "value": round(tg_tok_s, 2), "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "int8 dot (tg) / f16 MFMA on Q8-quantized activations (pp), f32 accumulate",
        "data": "synthetic",
        "config": {"workload": f"{wl.name}: {len(wl.all_mats())} MUL_MATs/pass, {wl.weight_bytes() / 1e9:.2f} GB quantized weights, "
                               f"pp{args.n_prompt} + tg{args.n_gen} per step; prompt pass with "
                               + ("logits for every token" if args.all_logits else "n_outputs = 1 as llama-bench runs it (last layer's FFN and the output projection on one row)"),
                   "n_prompt": args.n_prompt, "n_gen": args.n_gen, "n_outputs_pp": args.n_prompt if args.all_logits else 1,
                   "parallelism": "single GPU" if world == 1 else f"ggml row split over {world} GPUs, RCCL all-gather concat",
                   "tg_launch": "hipGraph replay" if graph is not None else "eager", "pp_launch": "hipGraph replay" if graph_pp is not None else "eager",
                   "weight_layout": "GGUF wire" if args.wire_layout else "planar rows for Q4_0 / Q8_0 / Q6_K (in-place repack at upload, SURVEY 8f-2), GGUF wire otherwise"},
        "tg128_tok_s": round(tg_tok_s, 2), "pp512_tok_s": round(pp_tok_s, 2),
        "tg_ms_per_token": round(tg_s / args.n_gen * 1e3, 4), "pp_ms_per_batch": round(pp_s * 1e3, 3),
        "tg_algo_GBs": round(wl.algo_bytes(1) / world / (tg_s / args.n_gen) / 1e9, 1),
        "pp_algo_TFLOPs": round(wl.flops(args.n_prompt, n_out_pp) / world / pp_s / 1e12, 1),
        "roofline": roof, "roofline_pp": roof_pp,
    }
    if rank == 0 and world == 1 and not args.no_e2e:
        lb = LlamaBench(wl.name)
        try:
            cores = min(len(os.sched_getaffinity(0)), 64)
            # the metric BASELINE.json names, end to end: the reference's unmodified libllama with this module loaded, all layers on
            # the device (attention, norms, KV cache, host-side graph build and sampling-free decode included)
            e2e = lb.run(True, min(cores, 16))
            if e2e is not None:
                out["e2e"] = e2e
            if not args.no_cpu_baseline:
                try:
                    out["cpu_baseline"] = cpu_baseline(wl, cores, lb)
                except Exception as e:          # the baseline is reported-only; never let it take the GPU numbers down
                    out["cpu_baseline"] = {"value": None, "unit": "tok/s (tg128)", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        finally:
            lb.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
