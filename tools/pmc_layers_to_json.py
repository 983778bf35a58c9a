"""Per-layer HBM rate of one CRAFT group from the two rocprofv3 --pmc passes of tools/prof_pages.py (FETCH_SIZE, WRITE_SIZE):
every convolution dispatch of the LAST step, in launch order, with its duration (kernel trace of the FETCH pass, i.e. under the
profiler), bytes fetched (KiB x 2: gfx950 tallies 128-B requests at 64 B) and written, and GB/s.  Merged into <out.json>["layers"].
  python tools/pmc_layers_to_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <pages> <out.json> <build>"""
import csv, json, sys

fetch_csv, write_csv, pages, out, build = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
CONV = ("conv3p", "conv3h", "conv1u", "gemm2_kernel", "gemm_sp_kernel", "conv1_direct", "conv1_split", "igemm_kernel", "conv3s")   # (detector-only runs: every such launch is CRAFT's)
LAYERS = ["slice1.0+slice1.3 +pool", "slice1.7", "slice1.10 (skip relu2_2 + pooled)", "slice2.14", "slice2.17 (skip relu3_2 + relu copy)", "slice3.20 +pool",
          "slice3.24", "slice3.27 (skip relu4_3 + pooled)", "slice4.30", "slice4.34", "slice4.37 (skip relu5_3 + relu copy)", "slice5.1 (dil 6)", "slice5.2",
          "upconv1.0", "upconv1.3", "upconv2.0", "upconv2.3", "upconv3.0", "upconv3.3", "upconv4.0", "upconv4.3", "conv_cls.0", "conv_cls.2", "conv_cls.4+.6+.8"]


# the f16x4 engine's 27 launches per group (nothing fused across layers but the pools and the ReLU copies)
LAYERS_X4 = ["slice1.0 (conv1_split, u8 canvas -> planes)", "slice1.3 +pool", "slice1.7", "slice1.10 (skip relu2_2 + pooled)", "slice2.14", "slice2.17 (skip relu3_2 + relu copy)",
             "slice3.20 +pool", "slice3.24", "slice3.27 (skip relu4_3 + relu copy)", "slice4.30 +pool", "slice4.34", "slice4.37 (skip relu5_3)", "slice5.1 (dil 6)", "slice5.2",
             "upconv1.0", "upconv1.3", "upconv2.0", "upconv2.3", "upconv3.0", "upconv3.3", "upconv4.0", "upconv4.3 (-> packed 32-channel rows)", "conv_cls.0 (packed pairs)",
             "conv_cls.2 (packed pairs)", "conv_cls.4 + .6 + .8 (packed pairs, fused tail)"]
LAYERS_X4_UNFUSED = LAYERS_X4[:-1] + ["conv_cls.4", "conv_cls.6 (fp32)", "conv_cls.8 (fp32)"]
# ... with upconv2.0 / 3.0 / 4.0 commuted with their upsamples (tuning key up_commute, the default): two launches each, no upsample kernels
LAYERS_X4_COMMUTED = []
for _l in LAYERS_X4:
    if _l in ("upconv2.0", "upconv3.0", "upconv4.0"):
        LAYERS_X4_COMMUTED += [_l + " (W_up . y at the low resolution -> fp32 z)", _l + " (skip half + upsample(z) in the epilogue)"]
    else:
        LAYERS_X4_COMMUTED.append(_l)


def rows(path, counter):
    r = [x for x in csv.DictReader(open(path)) if x["Counter_Name"] == counter and any(c in x["Kernel_Name"] for c in CONV)]
    r.sort(key=lambda x: int(x["Dispatch_Id"]))
    return r


f, w = rows(fetch_csv, "FETCH_SIZE"), rows(write_csv, "WRITE_SIZE")
n = len(LAYERS)
per_step = len(f) // 3 if len(f) % 3 == 0 else n
LAYERS_X4_FIRST_FUSED = ["slice1.0 + slice1.3 +pool (conv1_1 inside conv1_2's kernel, u8 canvas in)"] + LAYERS_X4_COMMUTED[2:]   # (round 6: conv3p.hip FIRST on pairs)
if f and per_step == len(LAYERS_X4_FIRST_FUSED) and "conv3p_kernel<64, 4, 2, true" in f[-per_step]["Kernel_Name"]:   # (its first launch is the fused pair)
    LAYERS = LAYERS_X4_FIRST_FUSED
    n = len(LAYERS)
elif f and "conv1_split" in f[-per_step]["Kernel_Name"]:       # the f16x4 engine's group (its first launch is conv1_split)
    LAYERS = LAYERS_X4_COMMUTED if per_step == len(LAYERS_X4_COMMUTED) else LAYERS_X4 if per_step == len(LAYERS_X4) else LAYERS_X4_UNFUSED
    n = len(LAYERS)
f, w = f[-per_step:], w[-per_step:]
# a 25th launch per step (an igemm fall-back for a thin layer) keeps its kernel name as label
layers = []
li = 0
for a, b in zip(f, w):
    us = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) * 1e-3
    fb, wb = 2 * 1024 * float(a["Counter_Value"]), 1024 * float(b["Counter_Value"])
    name = LAYERS[li] if li < n and per_step in (n, n + 1) else a["Kernel_Name"][:40]    # (a launch more than the list: the igemm fall-back of a thin layer keeps its kernel name)
    li += 1
    layers.append({"layer": name, "kernel": a["Kernel_Name"][:60], "us_under_profiler": round(us, 1), "fetch_MB": round(fb / 1e6, 1), "write_MB": round(wb / 1e6, 1),
                   "hbm_GB_per_s": round((fb + wb) / us / 1e3, 1), "frac_of_8TBps": round((fb + wb) / us / 1e3 / 8000.0, 3)})
j = json.load(open(out))
j["build"] = build
j["layers_note"] = "%d-page group, last of 3 steps, launch order; durations are those of the FETCH pass (profiled: ~5-10 %% slower than un-profiled)" % pages
j["layers"] = layers
json.dump(j, open(out, "w"), indent=1)
for l in layers:
    print("%-28s %-44s %8.1f us  %7.1f + %7.1f MB  %7.1f GB/s" % (l["layer"], l["kernel"][:44], l["us_under_profiler"], l["fetch_MB"], l["write_MB"], l["hbm_GB_per_s"]))
