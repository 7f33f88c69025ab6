#!/usr/bin/env python
"""bench.py -- audio-seconds/s of the Conformer-streaming encoder + CTC-greedy hot path.

Workload (BASELINE.json configs[1]): configs/conformer.yml, 32 synthetic
10 s utterances (1000 fbank frames x 80) PER GPU, random-init weights with the reference's
initialisers, V = 4233, ctc_greedy.  One "step" = one pass of the hot path over one batch:
features resident in HBM -> token ids + scores on device (+ one all-gather of the packed
hypotheses when N > 1).  Utterances are sharded across ranks (weak scaling, no data-path
collective other than that gather).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - dominant kernel's algorithmic FLOP/s vs the fp32-MFMA peak, duration measured
                 live with HIP events on the launch stream (ppasr_profile_* in the C-ABI)
  cpu_baseline - the torch-CPU oracle + numpy greedy timed on this host's cores (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
FRAME_SHIFT_S = 0.010


def conformer_flops_per_utt(T, F=80, d=256, ff=2048, L=12, k=15, V=4233):
    """Algorithmic FLOPs per utterance per kernel class (MAC = 2 FLOP), SURVEY.md §8(d).
    linear_pos(pos_emb) is weight-only and folded at create time -> not counted."""
    t1, f1 = (T - 1) // 2, (F - 1) // 2
    tp, f2 = (t1 - 1) // 2, (f1 - 1) // 2
    s1 = (4 * d * ff + 6 * d * d) * tp            # one k_ffn_qkv launch (FFN_macaron + QKV)
    s4 = (2 * k * d + 2 * d * d + 4 * d * ff) * tp  # one k_conv_ffn launch (dwconv + pw2 + FFN)
    per = {  # FLOPs per utterance of ONE launch of each kernel class
        "k_conv1": 2 * 9 * d * t1 * f1,
        "k_gemm_stream<conv2>": 2 * 9 * d * d * tp * f2,
        "k_gemm_stream<embed>": 2 * (d * f2) * d * tp,
        "k_ffn_qkv": s1,
        "k_attention": 6 * tp * tp * d,
        "k_out_glu": (2 * d * d + 4 * d * d) * tp,
        "k_attn_out_glu": 6 * tp * tp * d + (2 * d * d + 4 * d * d) * tp,  # attention + out-projection/GLU in one launch
        "k_conv_ffn": s4,
        "k_conv_ffn+ffn_qkv": s4 + s1,  # layer i's tail fused with layer i+1's head
        "k_ctc_head": 2 * d * V * tp,
    }
    return per, tp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (got {world})")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from ppasr_amd.model_utils.conformer.model import ConformerModel
    from ppasr_amd.parallel import gather_hypotheses
    from ppasr_amd.utils.synth import DEFAULT_VOCAB_SIZE, conformer_state_dict, synth_features

    V, L = DEFAULT_VOCAB_SIZE, 12
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=L, cnn_module_kernel=15)
    sd = conformer_state_dict(vocab_size=V, num_blocks=L, seed=1234)
    model = ConformerModel(80, V, streaming=True, encoder_conf=conf, state_dict=sd, device=device)
    B, T = args.batch, args.frames
    feats_np, lens_np = synth_features(B, T, seed=20240 + 200 + rank)
    feats = torch.from_numpy(feats_np).to(device)
    lens = torch.from_numpy(lens_np).to(device)

    def step():
        tokens, n_tokens, score = model.encode_greedy(feats, lens)
        if world > 1:
            return gather_hypotheses(tokens, n_tokens, score, dist)
        return tokens, n_tokens, score

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    audio_s_per_step = world * B * T * FRAME_SHIFT_S
    value = audio_s_per_step / (elapsed / args.steps)

    # ---- roofline leg: per-kernel durations from HIP events on the launch stream ----
    roofline = None
    kernels = {}
    if rank == 0:
        per_utt, tp = conformer_flops_per_utt(T, V=V, L=L)
        acc = {}
        reps = 3
        model.profile_kernels(True)
        for _ in range(reps):
            model.encode_greedy(feats, lens)
            for name, (ms, n) in model.read_kernel_profile().items():
                a = acc.setdefault(name, [0.0, 0])
                a[0] += ms
                a[1] += n
        model.profile_kernels(False)
        total_ms = sum(a[0] for a in acc.values()) / reps
        acc = {k: v for k, v in acc.items() if v[1] > 0}
        for name, (ms, n) in acc.items():
            flops_launch = per_utt[name] * B
            avg_ms = ms / n
            kernels[name] = {"launches_per_step": n // reps, "avg_ms": round(avg_ms, 4),
                             "tflops": round(flops_launch / (avg_ms * 1e-3) / 1e12, 2),
                             "share": round((ms / reps) / total_ms, 3)}
        dom = max(acc, key=lambda k: acc[k][0])
        ach = kernels[dom]["tflops"]
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (cannot be read live); the committed
        # figure is keyed by kernel class and dropped (null) when the dominant kernel has no entry.
        traffic = None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")) as f:
                traffic = json.load(f).get(dom, {}).get("hbm_bytes_per_launch")
        except OSError:
            pass
        roofline = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC)",
                    "avg_launch_ms": kernels[dom]["avg_ms"],
                    "whole_path_tflops_per_gpu": round(sum(per_utt[k] * acc[k][1] / reps for k in acc) * B
                                                       / (ms_per_step * 1e-3) / 1e12, 2),
                    "kernels": kernels}

    # ---- CPU baseline leg (rank 0, N=1 only): the oracle on this host's cores ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.conformer_oracle import ConformerOracle
        from oracle.ctc_decoders_oracle import greedy_tokens
        # the reference's own CPU configuration: InferencePredictor(num_threads=10)
        # (infer_utils/inference_predictor.py:20,68); more threads only oversubscribe these small GEMMs
        ncores = min(10, os.cpu_count() or 1)
        torch.set_num_threads(ncores)
        oracle = ConformerOracle(sd, num_blocks=L)
        xb, lb = feats_np[:2], lens_np[:2]
        oracle.get_encoder_out(xb, lb)  # warm-up
        done, t_cpu = 0, 0.0
        while t_cpu < 10.0:  # bounded sample: ~10 s of CPU work, cycling over the batch
            o = done % B
            xb, lb = feats_np[o:o + 2], lens_np[o:o + 2]
            t1 = time.perf_counter()
            probs = oracle.get_encoder_out(xb, lb).numpy()
            for p in probs:
                greedy_tokens(p)
            t_cpu += time.perf_counter() - t1
            done += len(xb)
        cpu = {"value": round(done * T * FRAME_SHIFT_S / t_cpu, 2), "unit": "audio-s/s", "cores": ncores, "kind": "port",
               "sample": f"{done} utterances of the same workload (batches of 2, cycling over the {B}), torch-CPU fp32 restatement of the Paddle "
                         f"reference + numpy greedy, {t_cpu:.1f} s of CPU work"}

    if rank == 0:
        line = {
            "metric": "audio-seconds/s (RTF^-1) Conformer-streaming fbank, batch32 per GPU, ctc_greedy",
            "value": round(value, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: Conformer streaming (configs/conformer.yml), fbank-80, "
                                   f"{B} x {T * FRAME_SHIFT_S:.0f} s utterances per GPU, V=4233, ctc_greedy, "
                                   "features resident in HBM -> token ids + scores on device",
                       "global_batch": world * B, "frames": T, "parallelism": f"utterance-dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
