"""profiles/r1_pmc/*_counter_collection.csv -> profiles/r1_traffic.json (read by bench.py)."""
import collections, csv, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC = os.path.join(ROOT, "profiles", "r1_pmc")


def per_launch(tag, counter):
    """{kernel short name: mean counter value per launch (KB)}"""
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(os.path.join(PMC, f"{tag}_counter_collection.csv"))):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void mmhn::", "")
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: tot[k] / n[k] for k in tot}


out = {"_comment": "HBM-side traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
       "profiles/r1_pmc/*.csv; scripts/pmc_kv2.sh = kv_only.py 20 20 64 6 [jacobi] [transpose], scripts/pmc_probe.sh = "
       "probe2.py 20 2048), k=20 fp64. FETCH_SIZE is doubled: on gfx950 it reports exactly 1/2 of these 8-B/lane streams "
       "(calibration cal_*: 744258 KB reported vs 1478656 KB actually read by 32768 single-tile problems); WRITE_SIZE is "
       "exact. Structurally-zero tiles (seed = 0, no PT == MT state: ~47 % of the tiles at k = 20) read nothing, which "
       "is why the products fetch less than one full vector."}
alg = {"kronvec": 2, "kronvec_T": 2, "jacobi_step": 4}
for name, tag, kern in (("kronvec", "kv", "k_sweep<double, false>"), ("kronvec_T", "kvt", "k_sweep<double, true>"),
                        ("jacobi_step", "js", "k_sweep<double, false>")):
    f, w = per_launch(tag + "_f", "FETCH_SIZE")[kern], per_launch(tag + "_w", "WRITE_SIZE")[kern]
    out[name] = {"fetch_kb": f, "write_kb": w, "bytes_per_launch": int((2 * f + w) * 1024),
                 "alg_bytes_per_launch": alg[name] * 64 * (2 ** 20) * 8}
P = 2048
ef, ew = per_launch("eval_f", "FETCH_SIZE"), per_launch("eval_w", "WRITE_SIZE")
ek = {"_comment": "per launch over 2048 n=20 full-k patients; bytes_per_patient = (2 x FETCH_SIZE + WRITE_SIZE) / patients"}
for name, kern in (("k_psolve_fwd", "k_psolve<double, false, true>"), ("k_psolve_adj", "k_psolve<double, true, true>"), ("k_pclass", "k_pclass<double>")):
    ek[name] = {"patients": P, "fetch_kb_reported": ef[kern], "write_kb": ew[kern],
                "bytes_per_patient": (2 * ef[kern] + ew[kern]) * 1024 / P}
out["eval_kernels"] = ek
json.dump(out, open(os.path.join(ROOT, "profiles", "r1_traffic.json"), "w"), indent=1)
for k, v in out.items():
    if k[0] != "_" and k != "eval_kernels":
        print(k, v["bytes_per_launch"] / v["alg_bytes_per_launch"])
print({k: round(v["bytes_per_patient"] / 1e6, 2) for k, v in ek.items() if k[0] != "_"})
