#!/usr/bin/env python3
"""bench.py -- frames/sec of the yolo-fastest-1.1 forward path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (net_input's conversion ... first conv ...
YOLO decode + NMS) over one batch of 64 synthetic 320x320 u8 BGR frames PER GPU
(`--input f32`: frames already converted to planar fp32, the
headline of rounds 1-3, reported beside the value either way), inputs resident in HBM,
ending with the NMS'd boxes of every frame of the job in host memory on rank 0
(RCCL gather of the fixed-size per-frame detection records for N > 1).  Weak
scaling by default: the global batch is 64*N.  `--global-batch 256` is BASELINE
config[4] as written (strong scaling): every step is the SAME 256 frames, cut
into contiguous shards of 256/N per GPU (ffcnn_amd.dist.shard_range).  Weights
are broadcast from rank 0 over RCCL once, before the timed region.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement"):
  value        whole-job frames/s
  roofline     the depthwise 3x3 kernel on BASELINE config[1] (320x320x64, batch 64):
               algorithmic bytes / mean launch time measured with HIP events on
               the launch stream, against the 8 TB/s HBM3E peak
  cpu_baseline the reference conv-v6.c path (oracle/_ref, built with the
               reference's own flags) timed on this box's host cores, N=1 only
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FRAMES_PER_GPU = int(os.environ.get("FFCNN_BENCH_FRAMES_PER_GPU", "64"))
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3
BF16_PEAK_TF = 2500.0             # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline figure includes 2:1 sparsity)


class DevBuf:
    """zero-copy torch view of a raw device pointer via __cuda_array_interface__"""

    def __init__(self, ptr, nbytes, typestr="|u1"):
        item = int(typestr[2:])
        self.__cuda_array_interface__ = {"shape": (nbytes // item,), "typestr": typestr, "data": (ptr, False), "version": 2}


def dev_tensor(torch, ptr, nbytes, typestr="|u1"):
    return torch.as_tensor(DevBuf(ptr, nbytes, typestr), device="cuda")


def cpu_flags():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def pick_ref_lib():
    """oracle/_ref build of the unmodified reference (conv-v6.c, reference flags) that this host can run."""
    ref = os.path.join(ROOT, "oracle", "_ref")
    have = cpu_flags()
    cands = []
    nat = os.path.join(ref, "native.cpuflags")
    if os.path.exists(nat) and set(open(nat).read().split()) <= have:
        cands.append(("v6_fast_native", "-Ofast -march=native (build host ISA)"))
    if {"avx2", "fma", "bmi2"} <= have:
        cands.append(("v6_fast_v3", "-Ofast -march=x86-64-v3"))
    cands.append(("v6_fast_sse2", "-Ofast -msse2"))
    for name, desc in cands:
        p = os.path.join(ref, "libffcnn_ref_%s.so" % name)
        if os.path.exists(p):
            return name, desc
    return None, None


CPU_WORKER = r"""
import os, sys, time
cpu = %(cpu)d
if cpu >= 0:
    try:
        os.sched_setaffinity(0, {cpu})          # one process per PHYSICAL core: its first hardware thread
    except (AttributeError, OSError):
        pass
sys.path.insert(0, %(root)r)
from oracle import orc
bgr, w, h = orc.load_bmp()
r = orc.Ref(%(variant)r)
for _ in range(3):
    r.set_input_image(bgr, w, h); r.forward()
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < %(secs)f:
    r.set_input_image(bgr, w, h); r.forward(); n += 1
dt = time.perf_counter() - t0
print(n, dt, r.n.bbox_num)
"""


def host_topology():
    """(hardware threads this process may run on, one representative thread per PHYSICAL core among them).
    Physical cores = distinct /sys/devices/system/cpu/cpuN/topology/thread_siblings_list values ("P = physical cores")."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores = {}
    for c in allowed:
        key = None
        for name in ("thread_siblings_list", "core_cpus_list"):
            try:
                key = open("/sys/devices/system/cpu/cpu%d/topology/%s" % (c, name)).read().strip()
                break
            except OSError:
                continue
        cores.setdefault(key if key is not None else "cpu%d" % c, []).append(c)
    return allowed, sorted(min(v) for v in cores.values())


def cpu_baseline(secs=8.0):
    """Reference CPU path on this box's host cores: 1 thread, then one process per PHYSICAL core (threads reported beside it),
    and the reference's own multi-threaded variant (conv-v4.c, the one OpenMP pragma of the repo: conv-v4.c:53) on as many threads."""
    variant, desc = pick_ref_lib()
    threads, reps = host_topology()
    ncore = len(reps)
    if variant is None:
        # no reference build travelled: time the oracle port instead (slower, scalar)
        from oracle import orc
        bgr, w, h = orc.load_bmp()
        o = orc.Oracle()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < secs:
            o.set_input_image(bgr, w, h)
            o.forward(1)
            n += 1
        dt = time.perf_counter() - t0
        o.close()
        return {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": "%d frames of test.bmp at 320x320 through oracle/ffcnn_oracle.c (-O2), 1 thread, %.1f s" % (n, dt)}

    def run(cpus, var=variant, env=None):
        procs = [subprocess.Popen([sys.executable, "-c", CPU_WORKER % dict(root=ROOT, variant=var, secs=secs, cpu=c)],
                                  stdout=subprocess.PIPE, text=True, env=env) for c in cpus]
        tot = 0.0
        frames = 0
        for p in procs:
            out = p.communicate()[0].split()
            if p.returncode == 0 and len(out) >= 2:
                tot += int(out[0]) / float(out[1])
                frames += int(out[0])
        return tot, frames

    one, f1 = run([-1])
    allc, fa = run(reps) if ncore > 1 else (one, f1)
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    out = {"value": round(one, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
           "sample": "%d frames (~%.0f s) of net_input+net_forward on test.bmp at 320x320, reference ffcnn.c+conv-v6.c "
                     "built %s, 1 thread" % (f1, secs, desc),
           "all_cores": {"value": round(allc, 3), "cores": ncore, "threads": len(threads),
                         "how": "%d independent processes (one NET each), one per physical core, each pinned to its core's first hardware thread; "
                                "%d hardware threads visible" % (ncore, len(threads))},
           "cpu": model}
    # the reference's own threaded variant: conv-v4.c with OMP_NUM_THREADS = physical cores (one process, one NET)
    v4 = "v4_fast_v3"
    if {"avx2", "fma", "bmi2"} <= cpu_flags() and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libffcnn_ref_%s.so" % v4)):
        try:
            # 8 threads, not one per core: conv-v4's only parallel region is one output-channel loop of a few microseconds, so every further thread
            # adds barrier time and nothing else (OMP_NUM_THREADS = 128 on the EPYC 9575F box: 1.3-2.1 frames/s in rounds 4 / 5 -- an over-subscription
            # artefact of the reference's pragma, not a property of the host; VERDICT r05 nit)
            nomp = min(ncore, 8)
            env = dict(os.environ, OMP_NUM_THREADS=str(nomp), OMP_PROC_BIND="close", OMP_PLACES="cores")
            r4, f4 = run([-1], var=v4, env=env)
            out["openmp_v4"] = {"value": round(r4, 3), "unit": "frames/s", "cores": nomp, "threads": nomp,
                                "how": "reference ffcnn.c+conv-v4.c (-Ofast -march=x86-64-v3 -fopenmp), one process, OMP_NUM_THREADS=%d (capped: the pragma's parallel "
                                       "region is a microsecond-sized channel loop, one thread per core of %d only adds barrier time), %d frames" % (nomp, ncore, f4)}
        except Exception as e:                                  # noqa: BLE001 -- an extra
            out["openmp_v4"] = {"error": repr(e)[:200]}
    return out


def kernel_roofline(torch, capi, stream):
    """BASELINE config[1]: depthwise 3x3 s1 p1, 64 channels x 64 frames of 320x320, leaky, BN folded."""
    N, Cc, H, W = 64, 64, 320, 320
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.rand((Cc * N, H, W), device="cuda", generator=g) * 2 - 1          # CNHW planes
    y = torch.empty_like(x)
    filt = torch.zeros((Cc, 16), device="cuda")
    filt[:, :9] = torch.rand((Cc, 9), device="cuda", generator=g) - 0.5
    filt[:, 12] = torch.rand((Cc,), device="cuda", generator=g) + 0.5
    filt[:, 13] = torch.rand((Cc,), device="cuda", generator=g) * 0.2 - 0.1
    torch.cuda.synchronize()
    alg_bytes = 2 * N * Cc * H * W * 4
    # The memory system needs ~40 ms of sustained traffic to settle after idle (profiles/r01_dw3_launch_series.txt: a
    # back-to-back series runs 560, 550, ..., 765 us around the 7th launch, then decays to a steady 550 us by the
    # 50th).  That transient is power management, not the kernel, so the timed region starts after 64 bare copies of
    # the same tensors (another kernel, so rocprofv3's per-kernel average of this command matches the number below).
    capi.diag().ffgpu_membench(y.data_ptr(), x.data_ptr(), alg_bytes // 2, 0, 1024, 64, stream.cuda_stream)
    us = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, Cc, Cc, 1, 1, 3, Cc, act=2,
                                 warmup=4, iters=50, stream=stream.cuda_stream)
    name = capi.kernel_name(N, W, H, Cc, Cc, 1, 1, 3, Cc)
    # context: what a bare 16-byte copy / read of the same bytes reaches on THIS GPU (ffgpu_membench)
    copy_us = min(capi.diag().ffgpu_membench(y.data_ptr(), x.data_ptr(), alg_bytes // 2, 0, b, 10, stream.cuda_stream) for b in (1024, 2048))
    read_us = capi.diag().ffgpu_membench(y.data_ptr(), x.data_ptr(), alg_bytes // 2, 2, 2048, 10, stream.cuda_stream)
    gbs = alg_bytes / (us * 1e-6) / 1e9
    traffic = traffic_source = None
    tf = os.path.join(ROOT, "profiles", "dw3x3_traffic.json")     # per-launch HBM bytes from the PMC passes
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            # NOT measured by this run: rocprofv3 PMC passes cannot run inside the timed process
            traffic_source = "profiles/dw3x3_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh, same kernel and shape; a cited constant, not collected by this run)"
        except (ValueError, OSError):
            traffic = None
    del x, y
    return {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "us_per_launch": round(us, 2),
            "algorithmic_bytes": alg_bytes, "workload": "dw3x3 s1 p1 320x320x64 batch 64 fp32 (BASELINE config[1])",
            "same_box_copy_GBs": round(alg_bytes / copy_us / 1e3, 1), "same_box_read_GBs": round(alg_bytes / 2 / read_us / 1e3, 1)}


def pw_roofline(torch, capi, stream):
    """BASELINE config[2]: pointwise 1x1 256->512 on 20x20, batch 256, leaky (MFMA-bound in fp32)."""
    N, ic, oc, H, W = 256, 256, 512, 20, 20
    g = torch.Generator(device="cuda").manual_seed(1235)
    x = torch.rand((ic * N, H, W), device="cuda", generator=g) * 2 - 1
    y = torch.empty((oc * N, H, W), device="cuda")
    filt = torch.zeros((oc, ic + 4), device="cuda")
    filt[:, :ic] = torch.rand((oc, ic), device="cuda", generator=g) - 0.5
    filt[:, ic] = torch.rand((oc,), device="cuda", generator=g) + 0.5
    filt[:, ic + 1] = torch.rand((oc,), device="cuda", generator=g) * 0.2 - 0.1
    torch.cuda.synchronize()
    # A compute-bound kernel reaches its sustained time only after ~25 ms of matrix work (tools: 290, 255, 243, 236, 231 us
    # for consecutive groups of 23 launches after the memory-bound measurement above; 228 us from the 100th on): clock /
    # power management, not the kernel.  The pre-heat therefore runs ANOTHER kernel of the same kind on the same tensors
    # (the streaming MFMA pointwise kernel), so that rocprofv3's per-kernel average of this command still is the average
    # of the launches timed here.
    capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2,
                            variant=capi.FFGPU.K_PW_MFMA, warmup=0, iters=80, stream=stream.cuda_stream)
    us = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2,
                                 warmup=4, iters=50, stream=stream.cuda_stream)
    flops = 2.0 * oc * ic * N * H * W
    tfs = flops / (us * 1e-6) / 1e12
    # beside it, RIGHT BEHIND the kernel's own launches and in front of the two cooler variants below (80 launches of pure bf16 MFMA leave the chip at its lowest
    # clock: measured last, they sat directly in front of the net's timed region and cost its first 20 steps 2-5 %): the MFMA-only floor of this launch on this box -- the launch's 4 915 200 v_mfma_f32_32x32x16_bf16 in k_pw_x3t's chunk pattern
    # on operands split from the same distributions and NOTHING else (lab kernel in libffcnn_hip_diag.so, wrong results by construction; tools/mfma_floor.py):
    # the time the power limit grants the arithmetic alone.  `frac` stays priced against the spec peak; `frac_of_floor` says how much of the gap is the chip's.
    floor = None
    if capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc) == "pw_x3t":
        try:
            import numpy as np
            D = capi.diag()
            rng = np.random.default_rng(1)

            def parts(v):
                v = v.astype(np.float32)
                out = []
                for _ in range(3):
                    t = (v.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
                    out.append((t.view(np.uint32) >> 16).astype(np.uint32))
                    v = v - t
                return out
            fr = np.zeros((18, 256, 4), np.uint32)
            wp, xp = parts(rng.uniform(-0.5, 0.5, (2, 256, 8))), parts(rng.uniform(-1, 1, (4, 256, 8)))
            for pt in range(3):
                for r in range(2):
                    fr[r * 3 + pt] = wp[pt][r, :, 0::2] | (wp[pt][r, :, 1::2] << 16)
                for j in range(4):
                    fr[6 + j * 3 + pt] = xp[pt][j, :, 0::2] | (xp[pt][j, :, 1::2] << 16)
            d_fr = torch.from_numpy(fr.view(np.int32)).cuda()
            n_mfma = int(round(6 * flops / 32768))
            us_floor = float(D.ffgpu_mfma_floor(d_fr.data_ptr(), n_mfma // (512 * 4 * 48), 512, 40, stream.cuda_stream))
            if us_floor > 0:
                floor = {"us_per_launch": round(us_floor, 2), "frac_of_floor": round(us_floor / us, 4), "GHz_at_full_matrix_pipe": round(n_mfma * 32 / 1024.0 / us_floor / 1e3, 3),
                         "what": "the launch's %d bf16 MFMAs alone on this box, two workgroups per CU, 128 accumulator registers, operands of the same distributions: "
                                 "no loads, no split, no LDS, no stores (ffgpu_mfma_floor, lab library); power-limited clock" % n_mfma}
        except (OSError, AttributeError, RuntimeError):
            floor = None
    # beside it: the fp32-MFMA kernel of rounds 2-3 on the same tensors (k_pw_gemm32: v_mfma_f32_32x32x2_f32, the kernel AUTO picked until round 4)
    f32k = None
    try:
        us32 = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2,
                                       variant=capi.FFGPU.K_PW_GEMM, warmup=4, iters=30, stream=stream.cuda_stream)
        f32k = {"kernel": "pw_gemm", "us_per_launch": round(us32, 2), "achieved": round(flops / us32 / 1e6, 2), "unit": "TFLOP/s", "frac": round(flops / us32 / 1e6 / FP32_MFMA_PEAK_TF, 4)}
    except RuntimeError:
        pass
    # beside it (NOT the product default, never used by the net's timed steps): the opt-in bf16-input variant of the same layer
    # (FFGPU_BF16_PW; its own tolerance in tests/test_gpu_kernels.py::test_pw_bf16) -- on the bf16 matrix cores the layer is
    # HBM-bound: 315 MB of fp32 input + output per launch
    bf = None
    try:
        us_bf = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, flags=capi.FFGPU.BF16_PW,
                                        variant=capi.FFGPU.K_AUTO, warmup=4, iters=30, stream=stream.cuda_stream)
        by = 4.0 * (ic + oc) * N * H * W
        bf = {"kernel": "pw_bf16", "us_per_launch": round(us_bf, 2), "bound": "hbm", "achieved": round(by / us_bf / 1e3, 1), "peak": HBM_PEAK_GBS,
              "unit": "GB/s", "frac": round(by / us_bf / 1e3 / HBM_PEAK_GBS, 4), "speedup_vs_f32": round(us / us_bf, 2),
              "note": "opt-in reduced precision (bf16 inputs, fp32 accumulation); tolerance 2^-7 scale' sum|w x| per output"}
    except RuntimeError:
        pass
    kname = capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc)
    split = kname in ("pw_x3", "pw_x3s", "pw_x3t")
    note = None
    if split:
        note = ("fp32-equivalent results from split operands: every fp32 value = three exact bf16 parts, six partial products per multiply-add on the bf16 "
                "matrix cores, fp32 accumulation (pw_x3t: ffgpu_pw_x3t.inc, tiled and double-buffered, v_mfma_f32_32x32x16_bf16; error as an fp32 summation "
                "order, tests/test_gpu_kernels.py::test_pw_x3t).  `peak` / `frac` price the kernel against what this form can reach: the bf16 dense "
                "peak / 6 products = 416.7 TFLOP/s of fp32-equivalent arithmetic (VERDICT r04 item 3); `frac_of_fp32_mfma_peak` is the same rate against "
                "the fp32 matrix peak the reference's arithmetic implies (157.3 TFLOP/s: the fp32 MFMAs run at the fp32 vector rate) and exceeds 1 by construction")
    peak = BF16_PEAK_TF / 6.0 if split else FP32_MFMA_PEAK_TF
    return {"bf16_opt_in": bf, "fp32_mfma_kernel": f32k, "bound": "mfma", "kernel": kname, "achieved": round(tfs, 2),
            "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tfs / peak, 4),
            "frac_of_fp32_mfma_peak": round(tfs / FP32_MFMA_PEAK_TF, 4), "frac_of_split_peak": round(tfs / (BF16_PEAK_TF / 6.0), 4) if split else None,
            "frac_base": "bf16 dense peak / 6 (split kernels, since round 5; rounds 1-4: the fp32 matrix peak = `frac_of_fp32_mfma_peak`)" if split else "fp32 matrix peak",
            "mfma_only_floor": floor, "note": note,
            "us_per_launch": round(us, 2), "dtype": "f32",
            "workload": "pw1x1 256->512 20x20 batch 256 fp32 (BASELINE config[2])"}


def launch_plan(gpus, env, ndev, node=False):
    """How `bench.py --gpus N` runs, decided from the command line, the environment and the visible devices alone:
         ("inline", None)    this process is the job (N = 1), or one rank of a torchrun job whose WORLD_SIZE == N
         ("node", None)      --node: the C node path, one process for all N devices
         ("torchrun", None)  N > 1 without a launcher: re-exec under torch.distributed.run, one rank per GPU
                             (FFCNN_BENCH_MULTI=node takes the C node path instead)
         ("refuse", why)     anything that would measure fewer than N GPUs and label it N (or label N GPUs as 1): rc != 0
       `--gpus N` uses N GPUs however it is launched, or fails loudly ("What's missing" 1)."""
    if gpus < 1:
        return "refuse", "--gpus %d" % gpus
    if node:
        if int(env.get("WORLD_SIZE", "1")) > 1:
            return "refuse", "--node is one process for all GPUs: do not launch it under torchrun"
        if ndev < gpus:
            return "refuse", "--gpus %d but %d HIP devices visible" % (gpus, ndev)
        return "node", None
    if "WORLD_SIZE" in env:
        try:
            world = int(env["WORLD_SIZE"])
        except ValueError:
            return "refuse", "WORLD_SIZE=%r" % env["WORLD_SIZE"]
        if world != gpus:
            return "refuse", "--gpus %d but WORLD_SIZE=%d: the line would be labelled with a GPU count the job does not have" % (gpus, world)
        if ndev < int(env.get("LOCAL_WORLD_SIZE", world)):
            return "refuse", "--gpus %d (WORLD_SIZE=%d) but %d HIP devices visible" % (gpus, world, ndev)
        return "inline", None
    if gpus == 1:
        return ("inline", None) if ndev >= 1 else ("refuse", "bench.py needs a HIP device (no CPU fallback)")
    if ndev < gpus:
        return "refuse", "--gpus %d but %d HIP device%s visible: refusing to run a %d-GPU job on fewer devices" % (gpus, ndev, "" if ndev == 1 else "s", gpus)
    return ("node", None) if env.get("FFCNN_BENCH_MULTI", "torchrun") == "node" else ("torchrun", None)


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def node_budget_s(ngpus, depth):
    """time limit of the `--node` child: creating an executor (arena + plan + one graph capture of 40 launches + first-use forward)
    takes ~0.35 s (measured: profiles/r04_exec_create.txt), a node holds ngpus x depth of them, RCCL's communicators ~1 s per device,
    torch + library start-up ~20 s, filling depth x ngpus input slots ~0.1 s each, the run itself < 5 s -- and a 3x margin on all of it"""
    return int(3 * (20 + 0.35 * ngpus * depth + 1.0 * ngpus + 0.1 * ngpus * depth + 5)) + 30


def node_line(args, ngpus):
    """`bench.py --node` in a child process (its own HIP context on all devices; a hang or crash there costs only this entry)."""
    env = {k: v for k, v in os.environ.items()
           if not (k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "GROUP_WORLD_SIZE",
                         "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS") or k.startswith("TORCHELASTIC") or k.startswith("TORCH_NCCL"))}
    # (at least 400 steps whatever --steps is: with 8 steps in flight a 20-step run would be mostly fill and drain)
    node_depth = 8 if ngpus == 1 else 4                         # executors: ngpus x depth (8 devices x 4 = 32 graph captures, 32 x 75 MB arenas)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--node", "--gpus", str(ngpus), "--steps", str(max(args.steps, 400)), "--warmup", str(max(args.warmup, 40)),
           "--depth", str(node_depth)]                         # (a host that waits for every step's records needs a deeper pipeline than the
                                                                #  enqueue-only loop above; N > 1: four steps in flight per device = the torchrun job's four chains)
    if args.global_batch > 0:
        cmd += ["--global-batch", str(args.global_batch)] + (["--merge-steps", str(args.merge_steps)] if args.merge_steps > 0 else [])
    try:
        # (bounded: at N > 1 this is the RCCL branch of the node path, which no box of rounds 1-3 could run; it must never cost the main line)
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=node_budget_s(ngpus, node_depth))
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
        d = json.loads(lines[-1])
        return {k: d.get(k) for k in ("value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "scaling", "host", "config")}
    except Exception as e:                                      # noqa: BLE001 -- an extra: never costs the main line
        return {"error": repr(e)[:400]}


def run_node(args):
    """The same job through the C node API (include/ffcnn_hip.h ffgpu_node_*), as a C host program would drive it: one process, one
    thread, N devices; per step ffgpu_node_submit (forward on every device, packed gather of the records over RCCL, one D2H) and
    ffgpu_node_wait (records of the whole step in host memory), `--depth` steps in flight.  torch only fills the input buffers."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("--node is one process for all GPUs: do not launch it under torchrun")
    import torch
    from ffcnn_amd import capi
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    N = args.gpus
    if torch.cuda.device_count() < N:
        raise SystemExit("--gpus %d but %d devices visible" % (N, torch.cuda.device_count()))
    strong = args.global_batch > 0
    G = args.global_batch if strong else FRAMES_PER_GPU * N
    D = max(1, min(args.depth, 8))
    # strong scaling: as in main(), a launch takes the shards of MS consecutive steps (a 32-frame launch leaves the small planes short
    # of waves): the node is created over MS * G frames, rank r's contiguous block = its shard of each of the MS steps behind each other
    MS = 1
    if strong:
        if G % N:
            raise SystemExit("--global-batch %d is not a multiple of %d GPUs" % (G, N))
        MS = args.merge_steps if args.merge_steps > 0 else max(1, 128 // (G // N))
    GL = MS * G                                                 # frames per launch (all devices)
    torch.cuda.set_device(0)
    capi.lib().ffgpu_set_device(0)
    os.environ.setdefault("FFGPU_BRANCH", "0" if D > 1 else "1")
    net = capi.Net()
    nd = capi.Node(net, N, GL, exec_flags=capi.FFGPU.CONCURRENT if D >= 3 else 0, node_flags=capi.Node.DEPTH(D))
    nd.set_scale(640, 320)
    rccl_ranks = nd.rccl_ranks()                                # communicators ncclCommInitAll really created (0 on one device)
    if N > 1 and rccl_ranks != N:
        raise SystemExit("--node --gpus %d: the node created %d RCCL communicators" % (N, rccl_ranks))
    # synthetic frames: the same global stream as the torchrun job (seed 1236, frame 0 = letterboxed test.bmp); every slot of
    # every device holds its own batch, resident in HBM before the timed region (D x 78.6 MB per device > the Infinity Cache)
    img = check = None
    try:
        bgr, w, h = capi.load_bmp(os.path.join(ROOT, "data", "test.bmp"))
        net.set_input_image(bgr, w, h)
        img = torch.from_numpy(net.input.copy())
        check = json.load(open(os.path.join(ROOT, "tests", "golden", "boxes.json")))["net_320x320_v0"]["boxes"]
    except Exception as e:                                      # noqa: BLE001
        print("bench: golden check unavailable: %r" % (e,), file=sys.stderr)
    fl = 3 * 320 * 320
    for slot in range(D):
        g = torch.Generator(device="cuda:0").manual_seed(1236 + slot)
        for r in range(N):
            lo, hi, dev = nd.shard(r)
            with torch.cuda.device(dev):
                view = torch.as_tensor(DevBuf(nd.input_slot_dev(r, slot), (hi - lo) * fl * 4, "<f4"), device="cuda:%d" % dev).view(hi - lo, 3, 320, 320)
                for c0 in range(lo, hi, 64):
                    cn = min(64, hi - c0)
                    chunk = torch.rand((cn, 3, 320, 320), device="cuda:0", generator=g)
                    view[c0 - lo:c0 - lo + cn] = chunk.to("cuda:%d" % dev)
                    del chunk
                if r == 0 and img is not None:
                    view[0] = img.to("cuda:%d" % dev)
    for d in range(N):
        torch.cuda.synchronize(d)
    recs = np.zeros(GL, capi.DETS_DTYPE)

    def run(nsteps):                                            # the pipelined loop itself runs in C (ffgpu_node_run), as a C host's would
        if nsteps > 0:
            nd.run(-(-nsteps // MS), recs)                      # launches (a ragged last one still does MS steps)

    pre = max(D, 128 // D * D)                                  # the device's sustained clock state (see main(): ~40 ms of forwards)
    run(pre)
    run(args.warmup)
    for d in range(N):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    run(args.steps)
    for d in range(N):
        torch.cuda.synchronize(d)
    dt = time.perf_counter() - t0
    # frame 0 of the LAST step sits in slot (steps - 1) % D ... every slot's frame 0 is the test image
    ok = None
    if check is not None:
        got = recs[0]["box"][: recs[0]["count"]]
        ok = bool(len(got) == len(check) and all(
            int(a["type"]) == int(b["type"]) and abs(float(a["score"]) - float(b["score"])) < 1e-4 and
            max(abs(float(a[k]) - float(b[k])) for k in ("x1", "y1", "x2", "y2")) < 0.05 for a, b in zip(got, check)))
    out = {"metric": "frames/sec yolo-fastest-1.1 @320x320 batch-64 per GPU (full forward: conv stack + YOLO decode + NMS, boxes in host memory)"
                     if not strong else "frames/sec yolo-fastest-1.1 @320x320 global batch %d sharded over the GPUs (full forward + boxes in host memory)" % G,
           "value": round(G * args.steps / dt, 1), "unit": "frames/s", "n_gpus": N, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "host": "C node API: one process, one host thread, ffgpu_node_run (= ffgpu_node_submit / ffgpu_node_wait, depth steps in flight) over include/ffcnn_hip.h",
           "config": {"workload": "yolo-fastest-1.1.cfg full net, 320x320x3 fp32 frames resident in HBM (BASELINE config[%d])" % (4 if strong else 3),
                      "input": "f32", "frames_per_gpu": G // N, "global_batch": G, "parallelism": "dp%d" % N, "steps_in_flight": D * MS,
                      "steps_per_launch": MS, "frames_per_launch": GL // N,
                      "exchange": "none (one device: the NMS kernel writes the records into pinned host memory)" if N == 1 else
                                  "ncclBroadcast of the weights at create; per step one grouped ncclSend/ncclRecv of the packed records per peer + one D2H",
                      "untimed_forwards_before_t0": pre + args.warmup,
                      "boxes_match_reference_golden_frame0": ok}}
    nd.close()
    net.close()
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-chain latency and the u8-input rate (N = 1 only, untimed extras)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the multi-GPU step (RCCL gather of the records on a side stream) even with one rank")
    ap.add_argument("--streams", type=int, default=4,
                    help="executors (one stream each) that take the batches in turn: up to that many batches are in flight")
    ap.add_argument("--input-sets", type=int, default=8, help="distinct synthetic batches resident in HBM, used in turn (8 x 78.6 MB does not fit the 256 MB Infinity Cache: every step reads its frames from HBM)")
    ap.add_argument("--split", action="store_true", help="FFGPU_SPLIT2 executors (two half-batch chains per batch)")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="multi-GPU: steps whose records travel in one RCCL gather (fewer, larger collectives); 0 = 64, "
                         "or less when --steps is small so that a short run still ships full groups")
    ap.add_argument("--merge-steps", type=int, default=0,
                    help="strong scaling: consecutive steps whose shards one launch processes together (0 = as many as make a "
                         "launch of ~128 frames: a 32-frame shard alone leaves the small-plane kernels short of waves)")
    ap.add_argument("--global-batch", type=int, default=int(os.environ.get("FFCNN_BENCH_GLOBAL_BATCH", "0")),
                    help="strong scaling (BASELINE config[4]: 256): a step is this many frames in total, cut into contiguous "
                         "shards of global/N per GPU; 0 = weak scaling, 64 frames per GPU")
    ap.add_argument("--node", action="store_true",
                    help="the C host path: ONE process drives all --gpus devices through ffgpu_node_create / submit / wait (RCCL broadcast of "
                         "the weights, packed gather of the records); not under torchrun")
    ap.add_argument("--depth", type=int, default=8, help="--node: steps in flight (FFGPU_NODE_DEPTH; one executor per slot and device)")
    ap.add_argument("--input", choices=["u8", "f32"], default=os.environ.get("FFCNN_BENCH_INPUT", "u8"),
                    help="what a step starts from, resident in HBM: u8 = 64 u8 BGR 320x320 images per GPU (SURVEY 8(d) row 4; the first kernel converts "
                         "them with net_input's arithmetic, ffcnn.c:259-289) -- the default since round 4; f32 = frames already converted to planar "
                         "fp32 (rounds 1-3's headline; reported beside the value either way)")
    ap.add_argument("--no-strong-extra", action="store_true", help="multi-GPU (or --force-gather) weak-scaling runs: skip the untimed strong_b256 extra (north_star's batch-256 job, "
                    "merged and one step per launch, on the same ranks)")
    ap.add_argument("--no-node-line", action="store_true", help="skip the extra c_node_api measurement (a child `bench.py --node` run by rank 0 after the timed job)")
    args = ap.parse_args()

    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    how, why = launch_plan(args.gpus, os.environ, ndev, node=args.node)
    if how == "refuse":
        print("bench.py: %s" % why, file=sys.stderr)
        raise SystemExit(2)
    if how == "node":
        args.node = True
        return run_node(args)
    if how == "torchrun":
        # `python bench.py --gpus N` with no launcher: become the launcher -- the same command line the driver uses for N > 1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        print("bench.py: --gpus %d without a launcher: re-exec as %s" % (args.gpus, " ".join(cmd[1:9])), file=sys.stderr)
        sys.stderr.flush()
        os.execv(sys.executable, cmd)

    import torch.distributed as dist
    from ffcnn_amd import capi
    from ffcnn_amd import dist as ffdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, (world, args.gpus)               # launch_plan() refused everything else
    torch.cuda.set_device(local)
    capi.lib().ffgpu_set_device(local)
    if world > 1 or args.force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    rccl_ranks = dist.get_world_size() if dist.is_initialized() else 0      # size of the communicator this job really created

    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit("--global-batch %d is not a multiple of %d GPUs (the gather moves equal-sized shards)" % (args.global_batch, world))
        lo, hi = ffdist.shard_range(args.global_batch, rank, world)      # this rank's contiguous frames of every step
        B = hi - lo
    else:
        lo, B = rank * FRAMES_PER_GPU, FRAMES_PER_GPU
    G = args.global_batch if strong else B * world
    # Strong scaling shrinks the shard with N (256 / 8 = 32 frames), and a 32-frame launch runs at 152 k frames/s per GPU
    # where a 64-frame one reaches 178 k and a 128-frame one 187 k (profiles/r02_b_batch_per_launch.txt): the last planes of the
    # net are 10x10 pixels, a launch needs frames to fill 256 CUs.  Steps are independent, so a launch takes the shards of
    # MS consecutive steps (MS x B frames, their records in step order): every step's boxes still reach rank 0 inside the
    # timed region, the schedule of the same work is coarser.  Weak scaling (the default) keeps MS = 1.
    MS = 1
    if strong:
        MS = args.merge_steps if args.merge_steps > 0 else max(1, 128 // B)
    # chain streams are PRIORITY streams: the runtime keeps a pool of (at most four) hardware queues per priority level, so the four
    # chains get queues of their own whatever other streams the process holds (torch's, RCCL's) -- at the default priority which chains
    # end up behind each other on one hardware queue depends on how many streams were created before them
    prios = [int(v) for v in os.environ.get("FFCNN_BENCH_STREAM_PRIORITY", "-1").split(",")]      # (a list: chain j takes prios[j % len])
    prio = prios[0]
    stream = torch.cuda.Stream(priority=prio)
    chain_streams = [stream]
    net = capi.Net()
    # weights: rank 0's folded filter rows -> every GPU over RCCL (one-off, outside the timed region)
    wptr, wbytes = net.weights_dev()
    if world > 1:
        wt = dev_tensor(torch, wptr, wbytes, "<f4")             # view of the library's device buffer
        wtmp = wt.clone() if rank == 0 else torch.zeros_like(wt)    # non-zero ranks start from zeros: the
        ffdist.broadcast_weights(dist, wtmp, src=0)                 # weights really arrive over RCCL
        wt.copy_(wtmp)
        torch.cuda.synchronize()
        net.weights_commit()                 # refresh the packed LDS images derived from the filter rows

    def job(MS, with_roofline, input_kind=None):
        """executors, inputs, rings, warm-up and the timed steps for MS steps per launch; returns what the report needs"""
        roof = roof_pw = None
        Bx = MS * B                                             # frames per launch
        # One executor per chain.  Single GPU: the NMS kernel writes the records straight into a pinned host mirror
        # (FFGPU_HOST_DETS), so the boxes are on the host when the step ends with no copy between two graph launches.
        # Several GPUs: the library's NMS kernel also writes each forward's records into a slot of a device ring
        # (ffgpu_exec_set_ring; nothing but graph launches sits on the compute stream); every --gather-every steps a side
        # stream gathers the finished group over RCCL (one message per rank and group instead of one per step: xGMI
        # collectives are latency-bound at this size) and moves the gathered block to rank 0's host while the next
        # forwards already run.  The boxes of a step reach rank 0 at most one group later, and the last, partial group is
        # flushed inside the timed region.
        gather_mode = world > 1 or args.force_gather
        host_dets = not gather_mode
        # Batches are independent, so consecutive steps go to S executors on S streams in turn (each its own arena and graph,
        # no events between them).  The chains run in lockstep (tools/ramp.py): S copies of every launch are on the device
        # together and fill the SIMDs that one launch leaves idle.  The head branch inside a chain is off.
        S = max(1, args.streams)
        M = args.gather_every
        if M <= 0:                                                  # groups a short run can fill at least once
            M = 64
            while M > S and 2 * M > max(-(-args.steps // MS), 2 * S):
                M //= 2
        M = max(M, (S + 1) // 2)
        if gather_mode and (2 * M) % S:
            raise SystemExit("--gather-every * 2 must be a multiple of --streams")
        os.environ.setdefault("FFGPU_BRANCH", "0" if S > 1 else "1")
        flags = (capi.FFGPU.HOST_DETS if host_dets else 0) | (capi.FFGPU.SPLIT2 if args.split else 0)
        if S >= 3:
            flags |= capi.FFGPU.CONCURRENT      # plan for throughput: several chains fill the device together
        exs = [net.executor(Bx, flags) for _ in range(S)]
        # the chain streams are made ONCE per process and every job of this run (the value, the strong_b256 and other-input extras) uses the same
        # ones: a fourth set of fresh priority streams lands two chains on one hardware queue (the fp32 extra read 150 k instead of 203 k behind
        # the two strong_b256 jobs; profiles/r05_b, r05_c)
        while len(chain_streams) < S:
            chain_streams.append(torch.cuda.Stream(priority=prios[len(chain_streams) % len(prios)]))
        streams = chain_streams[:S]
        ex = exs[0]
        model_bytes, model_flops = ex.work_model()

        # synthetic frames: the GLOBAL batch is seeded once (every rank draws the same stream and keeps its shard, so the strong-
        # scaling job processes the same 256 frames whatever N is); frame 0 of the job is the letterboxed test.bmp
        check = None
        K_in = max(1, args.input_sets)
        if (input_kind or args.input) == "u8":
            K_in *= 4                                               # 19.7 MB per batch: 32 sets (630 MB, as much as 8 fp32 sets) do not fit the 256 MB Infinity Cache
        xs = []
        g = torch.Generator(device="cuda").manual_seed(1236)
        img = None
        if rank == 0:
            # frame 0 = data/test.bmp through the library's own net_input; expected boxes are the
            # reference's (tests/golden/boxes.json, produced by the unmodified reference build)
            try:
                bgr, w, h = capi.load_bmp(os.path.join(ROOT, "data", "test.bmp"))
                net.set_input_image(bgr, w, h)
                img = torch.from_numpy(net.input.copy()).cuda()
                check = json.load(open(os.path.join(ROOT, "tests", "golden", "boxes.json")))["net_320x320_v0"]["boxes"]
            except Exception as e:
                print("bench: golden check unavailable: %r" % (e,), file=sys.stderr)
        # the steps take K distinct batches in turn (frame 0 is the test image in each of them, the rest differs): one batch used
        # over and over would sit in the 256 MB Infinity Cache and the first layer would never read HBM
        u8 = (input_kind or args.input) == "u8"
        img8 = None
        if img is not None:                                         # the same frame as u8 BGR: byte = round(255 x) (net_input computed x = byte * (1 / 255))
            img8 = torch.round(img * 255.0).clamp(0, 255).to(torch.uint8).flip(0).permute(1, 2, 0).reshape(320, 960).contiguous()
        for k in range(K_in):
            if u8:
                xk = torch.empty((Bx, 320, 960), dtype=torch.uint8, device="cuda")     # frame-major u8 BGR, pitch 960
            else:
                xk = torch.empty((Bx, 3, 320, 320), device="cuda")  # the shards of MS consecutive steps behind each other
            for q in range(MS):
                for c0 in range(0, G, 64):                          # one global batch in chunks; keep what falls into [lo, lo + B)
                    cn = min(64, G - c0)
                    if u8:
                        chunk = torch.randint(0, 256, (cn, 320, 960), dtype=torch.uint8, device="cuda", generator=g)
                    else:
                        chunk = torch.rand((cn, 3, 320, 320), device="cuda", generator=g)
                    a, b = max(c0, lo), min(c0 + cn, lo + B)
                    if a < b:
                        xk[q * B + a - lo:q * B + b - lo] = chunk[a - c0:b - c0]
                    del chunk
            if img is not None:
                xk[0] = img8 if u8 else img
            xs.append(xk)

        def fwd(e, xk, st):                                         # one step of executor e from resident frames xk on stream st
            if xk.dtype == torch.uint8:
                e.forward_bgr_dev(xk.data_ptr(), 320, 320, stream=st.cuda_stream)
            else:
                e.forward_dev(xk.data_ptr(), st.cuda_stream)
        x = xs[0]
        for e in exs:
            e.set_scale(640, 320)   # every frame is treated as a 640-wide source letterboxed to 320 (test.bmp's ratio)

        dptr, dbytes = ex.dets_dev()
        # ring of two groups of M slots: the library's NMS kernel writes forward k's records into slot k % 2M itself
        # (ffgpu_exec_set_ring), so nothing but graph launches sits on the compute stream
        ring = torch.empty((2, M, dbytes), dtype=torch.uint8, device="cuda") if gather_mode else None
        # what travels is the COMPACT form of a step's records (ffgpu_pack_records on the side stream: 25 KB instead of 198 KB
        # per step at batch 64 -- the fixed-size records are almost all unused box slots), room for 16 boxes per frame on average
        CAP = 16 * Bx
        pbytes = capi.packed_records_bytes(Bx, CAP)
        cring = torch.empty((2, M, pbytes), dtype=torch.uint8, device="cuda") if gather_mode else None
        big = torch.empty((world, M * pbytes), dtype=torch.uint8, device="cuda") if (gather_mode and rank == 0) else None
        glist = list(big.unbind(0)) if big is not None else None
        host = [torch.empty((world, M * pbytes), dtype=torch.uint8).pin_memory() for _ in range(2)] if (gather_mode and rank == 0) else None
        comm = torch.cuda.Stream() if gather_mode else None
        ev_comm = [torch.cuda.Event() for _ in range(2)]
        shipped = {"group": -1, "slot": 0}                          # where the newest step's records sit on rank 0's host

        ev_fwd = [[torch.cuda.Event() for _ in range(S)] for _ in range(2)]

        def ship(g, nslots=None):                                   # side stream: gather group g of the ring, D2H on rank 0
            ns = M if nslots is None else nslots                   # a flushed, partial group moves only the slots that were written
            nb = ns * pbytes
            for j in range(S):
                ev_fwd[g][j].record(streams[j])
            with torch.cuda.stream(comm):
                for j in range(S):
                    comm.wait_event(ev_fwd[g][j])                   # every chain has written its slots of the group
                capi.pack_records_dev(ring[g].data_ptr(), ns, Bx, Bx, CAP, cring[g].data_ptr(), comm.cuda_stream)
                ffdist.gather_records(dist, cring[g].view(-1)[:nb], dst=0, out=[t[:nb] for t in glist] if glist is not None else None)
                if rank == 0:
                    host[g][:, :nb].copy_(big[:, :nb], non_blocking=True)   # one D2H copy for the whole job's records of the group
                ev_comm[g].record(comm)

        def step(i, rev=False):
            j = i % S                                               # executor / stream of this step
            if rev:
                j = S - 1 - j                                       # (value_repeats: the chain that is otherwise enqueued last goes first)
            if not gather_mode:                                     # in-order streams: no events, no copies
                fwd(exs[j], xs[i % K_in], streams[j])
                shipped["group"] = j
                return
            g, slot = ffdist.ring_slot(i, M)
            if slot < S:
                streams[j].wait_event(ev_comm[g])                   # this group's previous gather has read it
            fwd(exs[j], xs[i % K_in], streams[j])
            shipped["group"], shipped["slot"] = g, slot
            if ffdist.group_due(i, M):
                ship(g)

        def flush(n_done):                                          # a partial last group still has to travel
            if gather_mode and n_done % M != 0:
                ship(ffdist.ring_slot(n_done, M)[0], n_done % M)

        def restart():                                              # forwards are counted from 0 again (slot 0 of group 0)
            if gather_mode:
                torch.cuda.synchronize()
                for j, e in enumerate(exs):                         # executor j's forward k is global step k * S + j
                    e.set_ring(ring.data_ptr() + j * dbytes, 2 * M // S, S * Bx)

        def fence():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # One HIP graph per executor, captured on its first forward and valid for every input buffer (the input pointer travels
        # through the executor's device parameter block): every (executor, input set) pair runs once here anyway, so nothing of
        # a first use -- graph capture, page mapping of a fresh buffer -- can land inside the timed region
        untimed = 0                                                 # forwards (launches of the whole net) enqueued before t0, reported in the line
        for k in range(max(K_in, S)):
            for j in range(S):
                fwd(exs[j], xs[k % K_in], streams[j])
                untimed += 1
        torch.cuda.synchronize()
        caps0 = [e.graph_captures for e in exs]                     # one graph per executor (+ one for the u8 form of the first kernel): nothing is captured after this point
        assert all(c <= 2 for c in caps0), caps0
        if gather_mode:                                             # ... and RCCL sets its communicator up on the first collective
            restart()
            with torch.cuda.stream(comm):
                ffdist.gather_records(dist, cring[0].view(-1), dst=0, out=glist)
            torch.cuda.synchronize()
        # The single-kernel rooflines (BASELINE config[1] / config[2]) are measured HERE, between set-up and the net's warm-up:
        # (a) after the executors exist -- the 3.4 GB + 0.3 GB these measurements allocate and free, taken first, left the
        # library's arenas on worse-placed memory (the same net then ran 15 % slower); (b) before the timed net -- the device
        # comes out of them in its sustained clock / memory state.  After idle it needs ~10 ms of work to get there
        # (tools/short_run.py: 20 steps straight after a 50 ms pause run at 153 k frames/s, the same 20 steps back to back at
        # 182 k), which a 5-step warm-up (1.8 ms) does not provide; both measurements are independent of the net's.
        if world == 1 and not args.no_kernel_roofline and with_roofline:
            roof = kernel_roofline(torch, capi, stream)
            roof_pw = pw_roofline(torch, capi, stream)
            torch.cuda.synchronize()
        else:
            # N > 1 (or --no-kernel-roofline): the same device state by other means -- ~40 ms of untimed forwards in front of the
            # W warm-up steps, on every rank.  Without them a short run at N > 1 is measured on a device that is still ramping
            # while the N = 1 run (which has just done its roofline launches) is not: 173 k against 184 k frames/s per GPU at 20
            # steps, a 6 % "scaling loss" that is measurement order and nothing else.
            for i in range(max(S, 128 // MS // S * S)):
                fwd(exs[i % S], xs[i % K_in], streams[i % S])
                untimed += 1
            torch.cuda.synchronize()
        # warm-up and timed steps are numbered from 0 each, so both start on a fresh group and end with a flush
        nl_warm, nl = -(-args.warmup // MS), -(-args.steps // MS)    # launches (a launch = MS steps; a ragged last one still does MS)
        restart()
        for i in range(nl_warm):
            step(i)
        untimed += nl_warm
        flush(nl_warm)
        fence()
        restart()
        fence()
        t0 = time.perf_counter()
        for i in range(nl):
            step(i)
        flush(nl)
        fence()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        assert [e.graph_captures for e in exs] == caps0, "a graph was captured inside the timed region"
        # value_repeats: the SAME K-step region five more times, back to back behind the value (untimed extras, `value` is the first
        # region and nothing else).  A 20-step region is 6.5 ms; DESIGN.md section 7 found one region in ten 7 % slower because the chain that was
        # enqueued last falls out of lockstep in its first round -- on the odd repeats that chain is enqueued FIRST, which tests the explanation.
        reps = []
        if world == 1 and not gather_mode and with_roofline:
            for r in range(5):
                fence()
                t1 = time.perf_counter()
                for i in range(nl):
                    step(i, rev=bool(r & 1))
                fence()
                reps.append(time.perf_counter() - t1)
        return dict(reps=reps, fwd=fwd, exs=exs, streams=streams, xs=xs, x=x, ex=ex, model_bytes=model_bytes, model_flops=model_flops, dt=dt, untimed=untimed, shipped=shipped, host=host, pbytes=pbytes, roof=roof, roof_pw=roof_pw, check=check, K_in=K_in, S=S, M=M, Bx=Bx, flags=flags, host_dets=host_dets, gather_mode=gather_mode)

    J = job(MS, True)
    exs = J["exs"]
    fwd = J["fwd"]
    streams = J["streams"]
    xs = J["xs"]
    x = J["x"]
    ex = J["ex"]
    model_bytes = J["model_bytes"]
    model_flops = J["model_flops"]
    dt = J["dt"]
    untimed = J["untimed"]
    shipped = J["shipped"]
    host = J["host"]
    pbytes = J["pbytes"]
    roof = J["roof"]
    roof_pw = J["roof_pw"]
    check = J["check"]
    K_in = J["K_in"]
    S = J["S"]
    M = J["M"]
    Bx = J["Bx"]
    flags = J["flags"]
    host_dets = J["host_dets"]
    gather_mode = J["gather_mode"]

    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        fps = G * args.steps / dt
        ok = ok_all = None
        if check is not None:
            if host_dets:
                rec = exs[shipped["group"]].dets_host()
            else:                                               # rank 0's block of the newest group, newest slot
                blk = host[shipped["group"]][0].numpy()[shipped["slot"] * pbytes:(shipped["slot"] + 1) * pbytes]
                rec = ffdist.unpack_records(blk, capi.DETS_DTYPE)
            # u8 input: the library scales the boxes as net_input does for the image it is given (ffcnn.c:267-273: a 320x320 image -> factor 1);
            # the golden boxes are in the pixels of the 640-wide source the frame was letterboxed from (factor 640 / 320)
            bs = 2.0 if args.input == "u8" else 1.0

            def golden(rec0):
                got = rec0["box"][: rec0["count"]]
                return bool(len(got) == len(check) and all(
                    int(a["type"]) == int(b["type"]) and abs(float(a["score"]) - float(b["score"])) < 1e-4 and
                    max(abs(bs * float(a[k]) - float(b[k])) for k in ("x1", "y1", "x2", "y2")) < 0.05 for a, b in zip(got, check)))
            ok = golden(rec[0])
            if host_dets:                                       # ... and the last step of EVERY chain (frame 0 of every input set is the test image)
                ok_all = all(golden(e.dets_host()[0]) for e in exs)
        per_gpu_s = dt / args.steps                             # seconds per step; every GPU handles B frames of it
        u8 = args.input == "u8"
        in_what = "u8 BGR frames -> net_input's conversion in the first kernel -> " if u8 else ""
        in_cfg = ("320x320 u8 BGR frames resident in HBM (SURVEY 8(d) row 4), converted by the first kernel with net_input's arithmetic (ffcnn.c:259-289)" if u8 else
                  "320x320x3 fp32 frames resident in HBM")
        if u8:                                                  # the first kernel reads 3 bytes per pixel instead of 12
            model_bytes = model_bytes - 9.0 * 320 * 320 * Bx
        out = {
            "metric": ("frames/sec yolo-fastest-1.1 @320x320 batch-64 per GPU (%sfull forward: conv stack + YOLO decode + NMS, boxes on rank 0)" % in_what)
                      if not strong else
                      "frames/sec yolo-fastest-1.1 @320x320 global batch %d sharded over the GPUs (%sfull forward + boxes on rank 0)" % (G, in_what),
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("yolo-fastest-1.1.cfg full net, %s (BASELINE config[3])" % in_cfg if not strong else
                                    "yolo-fastest-1.1.cfg full net, %s, global batch %d in contiguous shards (BASELINE config[4])" % (in_cfg, G)),
                       "input": args.input,
                       "frames_per_gpu": B, "global_batch": G, "parallelism": "dp%d" % world,
                       "steps_per_launch": MS, "frames_per_launch": Bx,
                       "input_sets": K_in,
                       "untimed_forwards_before_t0": untimed,
                       "untimed_kernel_launches_before_t0": ("~270 single-kernel launches of the config[1] / config[2] roofline measurements (~100 ms) between "
                                                             "set-up and the warm-up steps" if roof is not None else "none besides the forwards"),
                       "launches_per_step": ex.kernel_count, "arena_MB": round(ex.arena_bytes / 2**20, 1),
                       "executors": S, "pipelining": "%d executors on %d streams take the batches in turn%s" % (S, S, ", each split in two half-batch chains" if args.split else ""), "gather": ("RCCL gather of %d steps' records (packed: %d bytes per step and rank) + D2H on a side stream, overlapped with the next steps" % (M, pbytes)) if gather_mode else "records written to pinned host memory by the NMS kernel",
                       "graph_captures_per_executor": max(e.graph_captures for e in exs),     # (one per input format used: the u8 form of the first kernel has its own graph)
                       "weights": ("data/yolo-fastest-1.1.weights (broadcast from rank 0 over RCCL to %d ranks, untimed)" % rccl_ranks) if world > 1 else "data/yolo-fastest-1.1.weights (one GPU: loaded by net_load, no broadcast)",
                       "boxes_match_reference_golden_frame0": ok,
                       "boxes_match_reference_golden_frame0_every_chain": ok_all},
            # the whole net against the two ceilings that exist for it (per GPU): what the FUSED launch list must move
            # (each launch's inputs + outputs + filter rows once: ffgpu_exec_work_model) at the HBM peak, and the conv
            # stack's multiply-adds at the fp32 matrix peak.  Neither bounds the net tightly -- most launches are bound
            # by a wave's serial chain of MFMA + VALU issue -- but they are the honest denominators.
            "roofline_net": {"fused_algorithmic_bytes_per_batch": int(model_bytes / MS), "hbm_GBs": round(model_bytes / MS / per_gpu_s / 1e9, 1),
                             "hbm_frac": round(model_bytes / MS / per_gpu_s / 1e9 / HBM_PEAK_GBS, 4),
                             "flops_per_batch": int(model_flops / MS), "TFLOPs": round(model_flops / MS / per_gpu_s / 1e12, 2),
                             "mfma_f32_frac": round(model_flops / MS / per_gpu_s / 1e12 / FP32_MFMA_PEAK_TF, 4),
                             "batch": B},
        }
        # measured HBM traffic of one forward: a cited constant from the PMC passes of tools/net_traffic.sh (one chain, batch 64, u8 frames), not collected by this run
        try:
            nt = json.load(open(os.path.join(ROOT, "profiles", "r06_net_traffic.json")))
            if args.input == "u8" and B == 64:
                out["roofline_net"].update({"traffic": nt["hbm_bytes_per_forward"], "traffic_over_fused_model": nt["traffic_over_fused_model"],
                                            "traffic_source": "profiles/r06_net_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/net_traffic.sh over the %d launches of one 64-frame forward, "
                                                              "single chain, calibrated on a bare copy; a cited constant, not collected by this run)" % nt["launches_per_forward"]})
        except (OSError, KeyError, ValueError):
            pass
        if J["reps"]:
            rv = [G * args.steps / t for t in J["reps"]]
            srt = sorted(rv)
            out["value_repeats"] = {"n": len(rv), "unit": "frames/s", "min": round(srt[0], 1), "median": round(srt[len(srt) // 2], 1), "max": round(srt[-1], 1),
                                    "values": [round(v, 1) for v in rv], "median_over_value": round(srt[len(srt) // 2] / fps, 4),
                                    "order": ["chains enqueued 0..%d" % (S - 1) if not (r & 1) else "last chain first" for r in range(len(rv))],
                                    "what": "untimed extras: the same %d-step region %d more times back to back behind `value` (which is the FIRST region only); "
                                            "on the odd repeats the chain that is otherwise enqueued last goes first (DESIGN.md section 7)" % (args.steps, len(rv))}
        if roof is not None:
            out["roofline"] = roof
            out["roofline_pw"] = roof_pw
    # untimed extras (N = 1): one chain alone (latency of a batch), and the same job fed with u8 BGR frames (SURVEY 8(d) row 4:
    # "64 frames u8 BGR 320x320") through ffgpu_exec_forward_bgr_dev -- the letterbox / normalise kernel in front of the net
    if rank == 0 and world == 1 and not gather_mode and not args.no_extras:
        torch.cuda.synchronize()
        nlat = 30
        for i in range(5):
            fwd(exs[0], xs[i % K_in], streams[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(nlat):
            fwd(exs[0], xs[i % K_in], streams[0])
        torch.cuda.synchronize()
        out["roofline_net"]["single_chain_ms_per_batch"] = round((time.perf_counter() - t1) / nlat / MS * 1e3, 4)
        # ... and the same frames with TWO consecutive steps per launch (what the strong-scaling mode does with its shards): not the
        # reported value -- BASELINE's config is a 64-frame batch per step and launch --, the headroom a serving loop has if it may
        # put two batches into one launch
        if MS == 1 and K_in >= 2:
            ex2 = [net.executor(2 * Bx, flags) for _ in range(S)]
            x2 = [torch.cat([xs[(2 * k) % K_in], xs[(2 * k + 1) % K_in]]) for k in range(max(1, K_in // 2))]
            for e in ex2:
                e.set_scale(640, 320)
            n2 = max(S, 100 // S * S)
            for rep in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(n2):
                    fwd(ex2[i % S], x2[i % len(x2)], streams[i % S])
                torch.cuda.synchronize()
                t2 = time.perf_counter() - t1
            out["config"]["two_steps_per_launch"] = {"value": round(2 * Bx * n2 / t2, 1), "unit": "frames/s", "launches": n2,
                                                     "what": "untimed extra: 128 frames (two steps' batches) per launch on the same %d chains; NOT the reported configuration" % S}
            for e in ex2:
                e.close()
            del x2
    if rank == 0 and world == 1 and not gather_mode and not args.no_extras:
        # SURVEY 8(f) rows 2 / 3 as throughput (untimed extras): another darknet cfg (yolov3-tiny-shaped, implicit-GEMM kernels) and the
        # reference CLI's geometry, batch 16 per step on four chains (tools/other_nets.py)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import other_nets
            for e in exs:
                e.close()
            exs = []
            out["config"]["other_nets"] = other_nets.rows(torch, capi)
        except Exception as e:                                  # noqa: BLE001 -- an extra
            out["config"]["other_nets"] = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        for e in exs:
            e.close()
        exs = []
        out["cpu_baseline"] = cpu_baseline()
        out["gpu_vs_cpu_1thread"] = round(fps / out["cpu_baseline"]["value"], 1) if out["cpu_baseline"]["value"] else None
    for e in exs:
        e.close()
    exs = []
    if (gather_mode or not args.no_extras) and not strong and not args.no_strong_extra and 256 % world == 0:
        # ONE driver command, both scaling answers: beside the weak-scaling headline (64 frames per GPU) the same ranks -- at N = 1
        # the one GPU, whose figure is the denominator of north_star's ">= 7.5 x at 8 GPUs vs 1 GPU on batch 256" -- run
        # north_star's strong-scaling job -- every step is the SAME 256 frames, rank r takes shard_range(256, r, N) -- merged (as many consecutive
        # steps' shards per launch as make ~128 frames) and with one step per launch.  Untimed extra: own executors, inputs, warm-up, K timed steps,
        # every rank takes part (the gather is collective); `value` above is never replaced by it.
        xs = x = J = None
        torch.cuda.empty_cache()
        keep = (B, lo, G)
        G = 256
        lo, hi = ffdist.shard_range(G, rank, world)
        B = hi - lo
        MSs = max(1, 128 // B)
        sb = {"global_batch": G, "frames_per_gpu": B, "n_gpus": world, "rccl_ranks": rccl_ranks, "unit": "frames/s", "steps": args.steps,
              "what": "untimed extra: BASELINE config[4] / north_star's batch-256 job on the same %d rank(s), shards of %d frames; "
                      "merged = %d consecutive steps' shards per launch, unmerged = one step per launch" % (world, B, MSs)}
        for key, ms_ in (("merged", MSs), ("unmerged", 1)):
            if key == "unmerged" and MSs == 1:
                sb[key] = dict(sb["merged"])
                continue
            Jx = job(ms_, False)
            sb[key] = {"value": round(G * args.steps / Jx["dt"], 1), "ms_per_step": round(Jx["dt"] / args.steps * 1e3, 4), "steps_per_launch": ms_, "frames_per_launch": Jx["Bx"]}
            for e in Jx["exs"]:
                e.close()
            Jx = None
            torch.cuda.empty_cache()
        B, lo, G = keep
        if out is not None:
            out["strong_b256"] = sb
    if world == 1 and not gather_mode and not args.no_extras:
        # the same job from the OTHER input format, measured exactly like the value (own executors and inputs, same warm-up, K timed steps):
        # rounds 1-3 reported fp32-resident frames; since round 4 the value starts from the u8 images SURVEY 8(d) row 4 names
        xs = x = J = None
        torch.cuda.empty_cache()
        okind = "f32" if args.input == "u8" else "u8"
        J2 = job(MS, False, okind)
        out["config"]["fp32_resident_input" if okind == "f32" else "u8_bgr_input"] = {
            "value": round(G * args.steps / J2["dt"], 1), "unit": "frames/s", "steps": args.steps, "ms_per_step": round(J2["dt"] / args.steps * 1e3, 4),
            "what": ("same job, frames already converted to planar fp32 resident in HBM (78.6 MB per 64 frames instead of 19.7 MB): the headline of rounds 1-3" if okind == "f32" else
                     "same job from u8 BGR 320x320 frames resident in HBM -> ffgpu_exec_forward_bgr_dev (converted by the first kernel itself)")}
        for e in J2["exs"]:
            e.close()
        J2 = None
    if strong and MS > 1:
        # the same job with ONE step per launch (--merge-steps 1): what "batch 256 over N GPUs" gives when every launch carries exactly its
        # shard of one step.  Every rank takes part (the gather is collective); reported beside the merged headline, never instead of it.
        xs = x = J = None
        torch.cuda.empty_cache()
        J1 = job(1, False)
        if out is not None:
            out["strong_unmerged"] = {"value": round(G * args.steps / J1["dt"], 1), "unit": "frames/s", "ms_per_step": round(J1["dt"] / args.steps * 1e3, 4),
                                      "steps_per_launch": 1, "frames_per_launch": J1["Bx"],
                                      "what": "same job, --merge-steps 1: every launch carries this GPU's shard of ONE %d-frame step" % G}
            out["config"]["steps_per_launch_note"] = ("value = %d consecutive steps' shards per launch (~128 frames per launch at every N, so the launch size does "
                                                      "not shrink with N); strong_unmerged = one step per launch" % MS)
        for e in J1["exs"]:
            e.close()
        xs = x = J1 = None
    net.close()
    if world > 1 or args.force_gather:
        dist.destroy_process_group()
    if out is not None and not args.no_node_line:
        # north_star's host path -- plain C over the C-ABI, one process for all GPUs -- measured beside the torchrun job: rank 0
        # (alone by now: the process group is gone, the other ranks are leaving) runs `bench.py --node` on the same N devices
        xs = x = J = None
        torch.cuda.empty_cache()
        out["c_node_api"] = node_line(args, world)
        if world > 1 and not strong and not args.no_strong_extra and 256 % world == 0:      # ... and its batch-256 job (RCCL branch of the C node path, merged steps)
            import copy
            a256 = copy.copy(args)
            a256.global_batch, a256.merge_steps = 256, 0
            out["c_node_api_b256"] = node_line(a256, world)
    if out is not None:
        # the ONE JSON line is the last thing on stdout: RCCL's version banner sits in the C library's stdout buffer
        # until it is flushed
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
