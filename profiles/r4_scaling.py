#!/usr/bin/env python3
"""profiles/r4_scaling.json from the lines of profiles/r4_scaling.sh: per-rank step times of the N-way pixel-tile sharding rendered one rank at a
time on ONE GPU, and the speed-up they project.  REHEARSAL, UNMEASURED ON MULTI-GPU HARDWARE: what it contains is the load balance of the
tile deal and the cost of smaller launches; what it cannot contain is anything N GPUs do to each other (the RCCL reduce is priced, not run).
  projected speed-up(N) = full-frame step time / (max over ranks of the shard's step time + reduce time)
  reduce time = framebuffer bytes / 153 GB/s (one xGMI link: every peer reaches the root over its own link) -- and, as the pessimistic
  figure, a ring at half that rate moving (N - 1) / N of the buffer twice."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "gpurun_out", "r4_scaling")
LINK = 153e9


def load(tag):
    return json.load(open(os.path.join(D, tag + ".json")))


def config(name, full_tag, shard_fmt, ns, fb_bytes):
    full = load(full_tag)
    out = {"workload": full["config"]["workload"], "full_frame_ms_per_step": full["ms_per_step"], "full_frame_Mrays_per_s": full["value"],
           "framebuffer_bytes": fb_bytes, "ways": {}}
    for n in ns:
        ranks = [load(shard_fmt % (r, n)) for r in range(n)]
        ms = [x["ms_per_step"] for x in ranks]
        rays = [x["value"] * x["ms_per_step"] for x in ranks]           # Mrays/s x ms = k rays per step
        direct = fb_bytes / LINK * 1e3
        ring = 2.0 * fb_bytes * (n - 1) / n / (LINK / 2) * 1e3
        out["ways"][str(n)] = {
            "per_rank_ms_per_step": [round(v, 3) for v in ms], "max_ms": round(max(ms), 3), "mean_ms": round(sum(ms) / n, 3),
            "imbalance_max_over_mean": round(max(ms) / (sum(ms) / n), 4),
            "per_rank_share_of_rays": [round(v / sum(rays), 4) for v in rays],
            "sum_of_rank_rays_over_full_frame_rays": round(sum(rays) / (full["value"] * full["ms_per_step"]), 5),
            "per_gpu_rate_vs_full_frame": round((sum(rays) / n / (sum(ms) / n)) / full["value"], 4),
            "reduce_ms_direct_links": round(direct, 3), "reduce_ms_pessimistic_ring": round(ring, 3),
            "projected_speedup": round(full["ms_per_step"] / (max(ms) + direct), 3),
            "projected_speedup_pessimistic_reduce": round(full["ms_per_step"] / (max(ms) + ring), 3),
            "projected_efficiency": round(full["ms_per_step"] / (max(ms) + direct) / n, 4)}
    return out


def main():
    res = {"what": "REHEARSAL on one MI355X, unmeasured on multi-GPU hardware: bench.py --simulate-shard r/N for every rank, one gpurun call "
                   "(profiles/r4_scaling.sh); the N-GPU RCCL reduce is priced at the xGMI link rate, not run",
           "c4": config("c4", "c4_full", "c4_%d_of_%d", (2, 4, 8), 1920 * 1080 * 12),
           "c5": config("c5", "c5_full", "c5_%d_of_%d", (8,), 4096 * 4096 * 12)}
    json.dump(res, open(os.path.join(ROOT, "profiles", "r4_scaling.json"), "w"), indent=1)
    for k in ("c4", "c5"):
        for n, w in res[k]["ways"].items():
            print(k, "N =", n, "max %.1f mean %.1f ms  imbalance %.4f  projected speed-up %.2f (%.1f %%)" % (w["max_ms"], w["mean_ms"], w["imbalance_max_over_mean"], w["projected_speedup"], 100 * w["projected_efficiency"]))


if __name__ == "__main__":
    main()
