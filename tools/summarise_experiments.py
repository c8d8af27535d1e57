"""Measured tuning experiments (dropped and adopted) of one round, from the A/B timing logs (tools/ab_render.sh) and the counter
passes (tools/profile_experiments.sh):  python tools/summarise_experiments.py r03 > profiles/r03_scene_experiments.json
Per variant: best render time per configuration against the shipped library in the same run, and for the two kernels
that make up a render (extend: k_scene_persistent<RenderIO>, shade: k_shade) the vector-ALU busy fraction, the lanes
active per vector instruction and the instructions issued, over a 16 spp render of configs 2 and 5."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
CUS = 256
EXPERIMENTS = {
    "shadesort": {"log": "ab_shadesort.log", "flag": "-DPTX_SHADE_SORT=1",
                  "what": "k_shade regroups its block's 512 paths in LDS before shading them: scattering vertices first, one material kind after the other, then the cheap "
                          "paths (missed, met a lamp, last vertex) -- a stable counting sort on 8 keys (ballots, one prefix over waves x keys, one LDS exchange), no extra pass over "
                          "the path list; the survivors' list inherits the grouping (rays leaving surfaces of one kind are neighbours in the next extend round)"},
    "instsort": {"log": "ab_instsort.log", "flag": "-DPTX_SHADE_SORT=2",
                 "what": "the same in-block regrouping keyed by the instance the path's main ray hit (seven runs of the spatially ordered instance array, misses last): "
                         "rays leaving one part of the scene become neighbours in the survivors' list, hence in the next extend round"},
    "toplds0": {"log": "ab_toplds.log", "flag": "-DPTX_SCENE_TOP_LDS=16 -DPTX_SCENE_TOP_LDS_LEVEL=0",
                "what": "k_scene_persistent keeps the first 16 node records (breadth-first) of the instance-level tree in LDS (1 KB: what five 31 KB blocks per CU leave free)"},
    "toplds1": {"log": "ab_toplds.log", "flag": "-DPTX_SCENE_TOP_LDS=16 -DPTX_SCENE_TOP_LDS_LEVEL=1",
                "what": "the same 1 KB holds the first 16 node records of the mesh most instances refer to (the cow)"},
    "shadewaves6": {"log": "ab_shadewaves.log", "flag": "-DPTX_SHADE_WAVES=6", "variant": "sw6",
                    "what": "k_shade held to the 80 registers that let six waves share a SIMD instead of four (106 registers as the compiler leaves it): 45 spilled registers"},
    "shadewaves5": {"log": "ab_shadewaves.log", "flag": "-DPTX_SHADE_WAVES=5 -DPTX_SHADE_BLOCK=256", "variant": "sw5b256",
                    "what": "k_shade in 256-thread blocks held to 96 registers, five waves per SIMD: 19 spilled registers (sw5 in the same log: the same with 512-thread blocks, "
                            "whose waves come in pairs per SIMD, so four still)"},
    "ldsinsts": {"log": "ab_ldsinsts.log", "flag": "-DPTX_SHADE_LDS_INSTS=16", "variant": "li16",
                 "what": "k_shade copies the instance table of a scene with at most 16 instances into LDS (2.3 KB per block); the two matrices a scattering vertex reads -- seven 16-byte "
                         "gathers per lane, about 40 % of the kernel's vector-memory loads -- come from LDS instead (ds_read_b128 in the ISA): -0.4 % on config 2, nothing elsewhere; "
                         "the kernel is not held by the return path's throughput on those loads"},
    "shadeloads": {"log": "ab_shadeloads.log", "flag": "-DPTX_SHADE_EARLY_LOADS=2 -DPTX_SHADE_LDS_INSTS=16 (against 0 and 0)", "variant": "el2li",
                   "what": "k_shade asks for the main ray, its hit and the slot record together with the path's state, before it deals with the direct ray's result, so that the chains "
                           "of dependent loads overlap (el1: ray and hit only, el2: the slot record too), and reads the instance records of small scenes from LDS (ldsinsts above)"},
    "directlast": {"log": "ab_directlast.log", "flag": "(a patch, not kept: PTX_SHADE_DIRECT_LAST)", "variant": "dl",
                   "what": "k_shade asks for the previous direct ray's hit and weight at the top and adds what it brought after the vertex's work instead of before it (only paths that "
                           "are no camera rays have one, and their vertex adds nothing else, so the sum is the same: identical checksums): ten more live registers (106 -> 120), "
                           "nothing gained on the area-lit scenes, +5 % on the delta-lit room"},
    "pathblock": {"log": "ab_pathblock.log", "flag": "-DPTX_PATH_BLOCK=1024 (against 0: sample-major)", "variant": "pb1024",
                  "what": "which sample of which pixel a path of a pass is: 1,024 neighbouring pixels carry all the samples of the pass before the next 1,024 begin, so the few hundred "
                          "thousand paths in flight cover a patch of the film with all its samples instead of a third of the film with one (64 and 16,384 measured the same within 0.3 %)"},
    "chain": {"log": "ab_chain.log", "flag": "-DPTX_SCENE_CHAIN_MIN=8 -DPTX_SCENE_CHAIN_MIN_AFTER_LEAVES=24 (against one phase per iteration)", "variant": "chB",
              "what": "k_scene_persistent runs the enter, triangle and leave stages behind one another within one iteration when the stage before has made at least 8 lanes ready "
                      "(24 for leaving behind a triangle stage that was the iteration's own phase): a ray passes a mesh whose tree is a single leaf -- a wall of the room -- in one "
                      "iteration instead of three.  In the same log: ch65 = the restructured loop with chaining off (+0.5 to +1 %: its cost in registers), ch1/ch8/ch24 = one "
                      "threshold for both, chA = never leave behind an own triangle phase, chC = 4 and 40"},
    "stagethresholds": {"log": "ab_stages.log", "flag": "-DPTX_SCENE_T_ENTER_AFTER_WALK=24", "variant": "e24",
                        "what": "the chained stages generalised: any stage of the loop (walk, enter, triangles, leave, in this order) may run beside the iteration's phase when a "
                                "threshold of lanes is ready for it, at the start of the iteration (COLD) or behind an earlier stage (AFTER).  e24 = enter behind a walk stage with "
                                "24 lanes; in the same log w24/w32/w40 = walk whenever that many lanes can, whatever the phase (+2 to +4 %), w32e24 = both.  With the stages in the "
                                "order enter, walk, triangles, leave (profiles/r03_ab_stages_enterfirst.log) walking behind an enter stage gains 3 % on the instanced cows, but that "
                                "order costs 18 instead of 8 spilled registers and 2-4 % on the other scenes.  Nothing beyond enter -> triangles -> leave pays"},
    "crosscull": {"log": "ab_crosscull.log", "flag": "(a patch, not kept: PTX_SCENE_CROSS_CULL)", "variant": "cc",
                  "what": "k_scene_persistent bounds the object-space ray of an instance it enters by the closest hit found so far in other instances (object distance t comes back as "
                          "t * |T dir|, Trace::transform), so that subtrees and triangles beyond it are culled inside the instance as well, not only whole top-level leaves: identical "
                          "checksums, but the top-level culling has already removed nearly all of that work, and the bound's 25 instructions per entry and 3 per box step cost more"},
    "xcdparts": {"log": "ab_xcdparts.log", "flag": "-DPTX_SCENE_PARTS=8 (against 1: one claim counter)", "variant": "xp8",
                 "what": "k_scene_persistent cuts its item list into eight contiguous parts with a claim counter each; a block starts in part blockIdx % 8 (its XCD) and helps in the "
                         "next part when its own is dry, as k_mesh_persistent does: each XCD's L2 serves neighbouring rays"},
}


def ab_times(log):
    """{variant: {config: best seconds}} from an ab_render.sh log."""
    out, cur = {}, None
    path = os.path.join(ROOT, "profiles", f"{rnd}_{log}")   # the kept copy, if there is one
    if not os.path.exists(path):
        path = os.path.join(G, log)
    for line in open(path):
        m = re.match(r"== (\S+)", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"config (\S+): .*?: ([0-9.]+) s, .* checksum (\d+)", line)
        if m and cur:
            e = out.setdefault(cur, {}).setdefault(m.group(1), {"s": 1e9, "checksum": m.group(3)})
            e["s"] = min(e["s"], float(m.group(2)))
    return out


def kernel_counters(variant, leg):
    f = os.path.join(G, "prof_exp", f"{variant}_{leg}.json")
    if not os.path.exists(f):
        return {}
    agg = json.load(open(f)).get("counters", {})
    out = {}
    for name, label in (("k_scene_persistent<ptx::RenderIO, false, false", "extend"), ("k_shade<", "shade")):
        for k, m in agg.items():
            if name in k and "GRBM_GUI_ACTIVE" in m:
                cyc = m["GRBM_GUI_ACTIVE"] / 8.0
                out[label] = {"gpu_cycles_per_xcd": cyc, "valu_busy": round(4.0 * m["SQ_ACTIVE_INST_VALU"] / (cyc * CUS * 4), 4),
                              "lanes_per_valu_instruction": round(m["SQ_THREAD_CYCLES_VALU"] / m["SQ_INSTS_VALU"], 2), "valu_instructions": m["SQ_INSTS_VALU"], "waves": m["SQ_WAVES"]}
    return out


rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
result = {"round": rnd, "method": "tools/build_variant.sh NAME FLAGS; tools/ab_render.sh NAME (shipped and variant alternating on one box, best of 3 renders, twice); tools/profile_experiments.sh NAME "
                                  "(rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/prof_render.py 16 <leg>)",
          "configs": {"2": "refract_cow_cbox 1024x1024 256 spp depth 8", "5": "1,728 instanced cows 1024x1024 64 spp depth 8", "5f": "the same cows as one 10.1 M-triangle mesh, 1024x1024 64 spp depth 8", "1": "cbox 256x256 16 spp depth 4", "d": "room of instanced meshes and spheres lit by point, spot and directional lights, 1024x1024 64 spp depth 4"},
          "experiments": []}
for name, e in EXPERIMENTS.items():
    t = ab_times(e["log"])
    entry = {"name": name, "build_flags": e["flag"], "what": e["what"], "render_seconds": {}, "bit_identical_accumulator": True, "counters_16spp": {}}
    for cfg, ship in t.get("shipped", {}).items():
        var = t.get(e.get("variant", name), {}).get(cfg)
        if var:
            entry["render_seconds"][cfg] = {"shipped": ship["s"], "variant": var["s"], "change_percent": round(100.0 * (var["s"] / ship["s"] - 1.0), 1)}
            entry["bit_identical_accumulator"] &= var["checksum"] == ship["checksum"]
    for leg, cfg in (("refract_cow_cbox", "2"), ("instanced_cows", "5")):
        if kernel_counters(name, leg):
            entry["counters_16spp"][cfg] = {"shipped": kernel_counters("shipped", leg), "variant": kernel_counters(name, leg)}
    worse = [v["change_percent"] for v in entry["render_seconds"].values()]
    if worse and min(worse) > 0:
        entry["outcome"] = f"dropped: {min(worse):+.1f} % to {max(worse):+.1f} % render time"
    elif name in ("directlast", "crosscull"):
        entry["outcome"] = f"dropped: {min(worse):+.1f} % to {max(worse):+.1f} % render time"
    elif name == "ldsinsts":
        entry["outcome"] = f"adopted with shadeloads: {min(worse):+.1f} % to {max(worse):+.1f} % render time on its own (within run-to-run noise)"
    elif name == "stagethresholds":
        entry["outcome"] = f"dropped: {min(worse):+.1f} % to {max(worse):+.1f} % render time (the thresholds stay in the source, set to never)"
    elif worse and (max(worse) <= 0.1 or name == "chain"):
        entry["outcome"] = f"adopted: {min(worse):+.1f} % to {max(worse):+.1f} % render time"
    else:
        entry["outcome"] = "see numbers"
    result["experiments"].append(entry)
print(json.dumps(result, indent=1))
