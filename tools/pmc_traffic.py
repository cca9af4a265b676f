"""Fold the rocprofv3 PMC passes of tools/gpu_prof_round.sh (FETCH_SIZE / WRITE_SIZE summaries written by tools/rocpd_summary.py) into
profiles/pmc_traffic.json: HBM-side bytes per launch of the roofline kernel classes.  FETCH_SIZE is doubled as the gfx950 note of
MI355X_MICROARCH.md prescribes for wide coalesced reads (counter unit KB); WRITE_SIZE as reported.
usage: python tools/pmc_traffic.py <profiles dir> """
import json, os, re, sys

def parse(path):
    out, cur = {}, None
    for line in open(path):
        if line.startswith("# PMC"):
            cur = "pmc"; continue
        if cur != "pmc":
            continue
        if not line.startswith(" "):
            name = line.strip(); out[name] = {}
        else:
            m = re.match(r"\s+(\S+)\s+([0-9.]+)\s+over (\d+) dispatches", line)
            if m:
                out[name][m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out

def klass(d, tag, pat, src):
    f, w = parse(os.path.join(d, f"{tag}_pmc_FETCH_SIZE.txt")), parse(os.path.join(d, f"{tag}_pmc_WRITE_SIZE.txt"))
    tot_b, tot_n, rows = 0.0, 0, {}
    for name in f:
        if re.search(pat, name) and name in w:
            fs, n = f[name]["FETCH_SIZE"]; ws, _ = w[name]["WRITE_SIZE"]
            b = (2 * fs + ws) * 1024.0
            dur = f[name]["_duration_ns"][0]
            rows[name] = {"dispatches": n, "bytes_per_launch": round(b / n), "fetch_KB_raw_per_launch": round(fs / n, 1), "write_KB_per_launch": round(ws / n, 1),
                          "avg_us": round(dur / n / 1e3, 2), "TBps": round(b / dur / 1e3, 3)}
            tot_b += b; tot_n += n
    return {"bytes_per_launch": round(tot_b / max(tot_n, 1)), "dispatches": tot_n, "kernels": rows, "source": src}

def whole_forward(d, tag, forwards=3):
    """All kernels of the profiled run (tools/prof_forward.py: `forwards` UNet forwards + 3 prox applies + one pre_calculate, the latter two
    < 0.5 % of the bytes), per forward -- the figure the round-3 review summed by hand (49.7 GB at B = 16)."""
    f, w = parse(os.path.join(d, f"{tag}_pmc_FETCH_SIZE.txt")), parse(os.path.join(d, f"{tag}_pmc_WRITE_SIZE.txt"))
    fs = sum(v["FETCH_SIZE"][0] for v in f.values() if "FETCH_SIZE" in v)
    ws = sum(v["WRITE_SIZE"][0] for v in w.values() if "WRITE_SIZE" in v)
    by = {}
    for name in f:
        if name in w and "FETCH_SIZE" in f[name] and "WRITE_SIZE" in w[name]:
            key = re.sub(r"<.*", "", re.sub(r"^void ", "", name)).replace("dpir::", "")
            key = "act_split" if "act_split" in key else key
            by[key] = by.get(key, 0.0) + (2 * f[name]["FETCH_SIZE"][0] + w[name]["WRITE_SIZE"][0]) * 1024.0 / forwards
    top = dict(sorted(((k, round(v / 1e9, 2)) for k, v in by.items()), key=lambda kv: -kv[1])[:8])
    return {"GB_per_forward": round((2 * fs + ws) * 1024.0 / forwards / 1e9, 2), "fetch_GB_raw": round(fs * 1024.0 / forwards / 1e9, 2),
            "write_GB": round(ws * 1024.0 / forwards / 1e9, 2), "GB_per_forward_by_kernel_family": top}


d = sys.argv[1]
src = f"{d}/<case>_pmc_{{FETCH,WRITE}}_SIZE.txt: separate rocprofv3 --pmc passes over tools/prof_forward.py (tools/gpu_prof_round.sh), FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, KB -> bytes, launch-weighted over the kernels of the class"
out = {
    "ffhq_B16_256_f16x3": klass(d, "ffhq_f16x3", r"conv7_mfma_kernel|conv6_mfma_kernel|conv8_fused_kernel", src),
    "ffhq_B16_256_f32": klass(d, "ffhq_f32", r"conv2_mfma_kernel<3|conv2_mfma_kernel<1, 8|conv_mfma_kernel<3", src),
    "imagenet256_B32_256_f16x3": klass(d, "in256_f16x3", r"conv7_mfma_kernel|conv6_mfma_kernel|conv8_fused_kernel", src),
    "fftprox_sf1_B16_256": klass(d, "ffhq_f16x3", r"rfft4_rows_kernel|cfft4_cols_kernel<2|irfft4_rows_kernel|rfft_rows_kernel|cfft_cols_kernel<16, 16, 2|cfft_cols_kernel<16, 2|irfft_rows_kernel", src),
    "fftprox_sf4_B32_256": klass(d, "in256_f16x3", r"rfft4_rows_kernel|cfft4_cols_kernel<3|irfft4_rows_kernel|rfft_rows_kernel|cfft_cols_kernel<16, 16, 3|cfft_cols_kernel<16, 3|irfft_rows_kernel", src),
}
if os.path.exists(os.path.join(d, "prox512_pmc_FETCH_SIZE.txt")):
    out["fftprox_sf4_B8_512"] = klass(d, "prox512", r"rfft4_rows_kernel|cfft4_cols_kernel<3|irfft4_rows_kernel", src)
    out["fftprox_sf4_B8_512"]["note"] = "configs[4]'s data step alone (tools/prof_forward.py with PROF_UNET=0 PROF_SIZE=512 PROF_B=8 PROF_SF=4)"
out["ffhq_B16_256_f16x3"]["whole_forward"] = whole_forward(d, "ffhq_f16x3")
out["imagenet256_B32_256_f16x3"]["whole_forward"] = whole_forward(d, "in256_f16x3")
for k in ("fftprox_sf1_B16_256", "fftprox_sf4_B32_256"):
    out[k]["note"] = "per-launch average over the three kernels of one apply AND the pre_calculate launches of the profiled run; bytes per apply = sum of the three apply kernels' rows"
json.dump(out, open(os.path.join(os.path.dirname(d.rstrip('/')), "pmc_traffic.json"), "w"), indent=1)
print("whole forward:", out["ffhq_B16_256_f16x3"]["whole_forward"], out["imagenet256_B32_256_f16x3"]["whole_forward"])
for k, v in out.items():
    print(k, v["bytes_per_launch"], v["dispatches"], {n[:60]: (r["bytes_per_launch"], r["avg_us"], r["TBps"]) for n, r in v["kernels"].items()})
