#!/usr/bin/env python3
"""VALU instruction mix of k_trace_coop<false, 4, false> and the mean time one of its VALU instructions holds a SIMD.

Why: the kernel is limited by instruction issue, and the hardware counters say how MANY VALU instructions a launch
executes (SQ_INSTS_VALU) but not how long they hold the SIMD -- SQ_ACTIVE_INST_VALU counts every VALU instruction as one quad-cycle on
gfx950, whatever it is.  profiles/valu_rate2.hip measured the issue time per instruction class on this chip (profiles/*/valu_rate.json,
ns per wave instruction and SIMD at 8 waves per SIMD: add / mul / fma / logic 1.04-1.25, min / max / compare / select / DPP / 3-operand
integer 1.7-1.85, rcp / sqrt 3.4).  This script classifies the kernel's own instructions and weighs them by how often their loop runs:

    mean_issue_ns = sum_i n_i * c_i / sum_i n_i          n_i: dynamic count of class i (static count per block x executions of the block)
    issue fraction = (SQ_INSTS_VALU per launch / 1024 SIMDs) * mean_issue_ns / launch time                  (profiles/summarize.py)

Block executions come from the counting variant of the kernel (bench.py roofline.wave_occupancy: node_phase_iters, leaf_phase_iters,
wave_iters): the inner pop + node loop runs node_phase_iters (+ wave_iters for the pop that ends it), everything else in the outer
loop about once per wave iteration (the leaf phase runs in 99 % of them on C4).  The prediction VALU / ray is printed next to the
measured one as a check of the weights.
usage: python profiles/valu_mix.py <bench line json with roofline.wave_occupancy> [valu_rate.json] > profiles/<tag>/valu_mix.json"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("-fno-slp-vectorize -DART_COOP_WAVES_PER_SIMD=8 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "
         "-fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero").split()      # = ada-ray-tracer_amd/Makefile
KERNEL = "_ZN3art12k_trace_coopILb0ELi4ELb0ELb0EEEvPKNS_8DevSceneENS_9TraceArgsE"      # <STATS = false, G = 4, OVF = false, INST = false>

CLASS = [   # (regex on the mnemonic, key in valu_rate.json)
    (r"v_fma_f32|v_fmac_f32", "fma"), (r"v_mul_f32", "mul"), (r"v_add_f32", "add"), (r"v_sub(rev)?_f32", "sub"),
    (r"v_(min|max)3_f32", "max3"), (r"v_med3", "med3"), (r"v_(min|max)_f32", "max"),
    (r"v_cmp\w*_[fiu]\d+_e64|v_cmp\w*_e64", "cmp_sgpr"), (r"v_cmp", "cmp_vcc"), (r"v_cndmask", "cndmask_sgpr"),
    (r"v_mov_b32_dpp", "mov_dpp"), (r"v_(add|sub)(_co)?_u32_dpp", "add_u32_dpp"), (r"v_min_u32_dpp", "min_u32_dpp"), (r"v_\w+_f32_dpp", "add_f32_dpp"),
    (r"v_subb?rev_co_u32|v_subb_co|v_addc_co", "subbrev"), (r"v_perm_b32", "perm"), (r"v_cvt_f32_ubyte|v_cvt_f32_u32|v_cvt", "cvt_f32_u32"),
    (r"v_and_or_b32", "and_or"), (r"v_lshl_or", "lshl_or"), (r"v_lshl_add", "lshl_add"), (r"v_add3", "add3"), (r"v_xor3|v_xad", "xor3"),
    (r"v_lshlrev|v_lshrrev|v_ashrrev", "lshlrev"), (r"v_(add|sub|subrev)(_co)?_u32", "add_u32"), (r"v_(and|or|xor|not)_b32", "and_b32"),
    (r"v_min_u32|v_max_u32|v_min_i32|v_max_i32", "min_u32"), (r"v_bfe|v_bfi", "bfe"), (r"v_mul_lo|v_mul_hi|v_mad_u64", "mul_lo"), (r"v_mad_[ui]", "mad_u24"),
    (r"v_rcp", "rcp"), (r"v_sqrt|v_rsq", "sqrt"), (r"v_div_scale", "div_scale"), (r"v_div_fmas", "div_fmas"), (r"v_div_fixup", "div_fixup"),
    (r"v_pk_fma", "pk_fma"), (r"v_pk_mul", "pk_mul_f32"), (r"v_pk_add", "pk_add_f32"), (r"v_fma_f64|v_mul_f64|v_add_f64", "fma_f64"),
    (r"v_mov_b32|v_readfirstlane|v_readlane|v_accvgpr|v_swap", "mov"),
]


def classify(mn):
    for rx, key in CLASS:
        if re.match(rx, mn):
            return key
    return None


def main():
    bench = json.load(open(sys.argv[1]))
    rates = json.load(open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r2_final", "valu_rate.json")))["ns_per_wave_instruction_per_simd"]
    occ = bench["roofline"]["wave_occupancy"]
    asm = "/tmp/art_kernels_mix.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S",
                                                              os.path.join(ROOT, "ada-ray-tracer_amd", "csrc", "art_kernels.hip"), "-o", asm], stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    # basic blocks with LLVM's loop annotations ("in Loop: Header=BBn_m Depth=d")
    blocks = []; cur = {"label": "entry", "depth": 0, "header": None, "ins": []}
    for l in body[1:]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l) or re.match(r"^; %bb\.\d+:\s*(;.*)?$", l)
        if m:
            blocks.append(cur)
            note = l[l.index(";"):] if ";" in l else ""
            d = re.search(r"Depth=(\d+)", note); h = re.search(r"Header=(BB\d+_\d+)", note) or re.search(r"Loop Header", note)
            cur = {"label": l.split(":")[0].strip("; %"), "depth": int(d.group(1)) if d else 0, "header": (h.group(1) if h and h.groups() else None), "note": note, "ins": []}
            continue
        if re.match(r"^\s+;\s+=>", l) or re.match(r"^\s+;\s+(Parent|Child)", l):
            d = re.search(r"Depth=(\d+)", l)
            if "Inner Loop Header" in l and d:
                cur["depth"] = int(d.group(1)); cur["inner_header"] = True
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur["ins"].append(t.split()[0])
    blocks.append(cur)
    # the node loop = the depth-2 loop that contains s_setprio (pop + node step); its blocks carry that header's name
    node_header = None
    for i, b in enumerate(blocks):
        if "s_setprio" in b["ins"] and b["depth"] == 2:
            node_header = b["header"] or b["label"].lstrip(".L")
    in_node = lambda b: b["depth"] == 2 and (b["header"] == node_header or b["label"].lstrip(".L") == node_header)
    # inside the node loop, the blocks before s_setprio's block in layout order are the pop (they also run once more when the loop is left)
    node_blocks = [b for b in blocks if in_node(b)]
    w_node = float(occ["node_phase_iters"]); w_outer = float(occ["wave_iters"]); w_leaf = float(occ["leaf_phase_iters"])
    def weigh(outer_scale):
        return _weigh(blocks, in_node, rates, w_node, w_outer * outer_scale, w_leaf, w_outer)
    counts, unknown, dyn_total, static = weigh(1.0)
    lo = weigh(0.0); hi = weigh(1.0)
    mean_of = lambda r: sum(n * rates[k] for k, n in r[0].items()) / max(r[2], 1.0)
    mean_ns = mean_of((counts, unknown, dyn_total, static))
    out = {"kernel": "k_trace_coop<false, 4, false>", "static_valu_instructions": static,
           "block_weights": {"node_phase_iters": w_node, "leaf_phase_iters": w_leaf, "wave_iters": w_outer},
           "dynamic_mix_fraction": {k: round(n / dyn_total, 4) for k, n in sorted(counts.items(), key=lambda kv: -kv[1])},
           "mean_issue_ns_per_valu_instruction": round(mean_ns, 4),
           "mean_issue_ns_if_the_conditional_outer_code_never_ran": round(mean_of(lo), 4),
           "note": "refill / retire code of the outer loop is conditional; weighing it once per wave iteration over-counts it (the prediction below is "
                   "an upper estimate of SQ_INSTS_VALU), but the mean issue time moves by < 1 % between 'never' and 'always'",
           "predicted_valu_wave_instructions_of_the_counting_pass": dyn_total,
           "unclassified_mnemonics_priced_as_max": unknown,
           "rates_from": os.path.relpath(sys.argv[2], ROOT) if len(sys.argv) > 2 else "profiles/r2_final/valu_rate.json"}
    json.dump(out, sys.stdout, indent=1)
    print()


def _weigh(blocks, in_node, rates, w_node, w_outer, w_leaf, w_pop_extra):
    counts = {}; unknown = {}
    dyn_total = 0.0; static = {"node_loop": 0, "outer": 0, "leaf": 0, "other": 0}
    for b in blocks:
        valu = [m for m in b["ins"] if m.startswith("v_")]
        if not valu:
            continue
        if in_node(b):
            has_step = "s_setprio" in b["ins"] or any(m.startswith("ds_write") for m in b["ins"])
            w = w_node if has_step else (w_node + w_pop_extra); static["node_loop"] += len(valu)
        elif b["depth"] >= 1:
            leafy = any(m.startswith("v_div_") or m.startswith("v_rcp") for m in valu)
            w = w_leaf if leafy else w_outer; static["leaf" if leafy else "outer"] += len(valu)
        else:
            w = 0.0; static["other"] += len(valu)          # prologue / epilogue: once per wave
        for m in valu:
            k = classify(m)
            if k is None or k not in rates:
                unknown[m] = unknown.get(m, 0) + 1; k = "max"        # unknown mnemonics are priced like the compare / select class
            counts[k] = counts.get(k, 0.0) + w
            dyn_total += w
    return counts, unknown, dyn_total, static


if __name__ == "__main__":
    main()
