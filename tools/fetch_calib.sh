#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on K3's access patterns (tools/fetch_calib.hip): counter KB per kernel against known bytes.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fetch_calib
mkdir -p "$OUT" "$ROOT/tools/_build"
[ -x "$ROOT/tools/_build/fetch_calib" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 "$ROOT/tools/fetch_calib.hip" -o "$ROOT/tools/_build/fetch_calib" || exit 1
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/$ctr" -o run -- "$ROOT/tools/_build/fetch_calib" > "$OUT/$ctr.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, json
out = sys.argv[1]
known = {"calib_stream16": 3 << 30, "calib_nodes12": (3 << 30) // 12 // 1024 * 1024 * 12, "calib_store12": (3 << 30) // 12 * 12}
nq = (3 << 30) // 16 // 4
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if row["Counter_Name"] == ctr:
                k = row["Kernel_Name"].split("(")[0].strip()
                res.setdefault(k, {})[ctr] = res.get(k, {}).get(ctr, 0.0) + float(row["Counter_Value"]) * 1024.0
doc = {"method": "tools/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); 3 GiB per pattern, far beyond the 256 MB Infinity Cache",
       "patterns": {}}
for k, v in sorted(res.items()):
    e = {"FETCH_SIZE_bytes": v.get("FETCH_SIZE"), "WRITE_SIZE_bytes": v.get("WRITE_SIZE")}
    if k in known:
        e["known_bytes"] = known[k]
        if k != "calib_store12" and v.get("FETCH_SIZE"): e["fetch_over_known"] = round(v["FETCH_SIZE"] / known[k], 4)
        if k == "calib_store12" and v.get("WRITE_SIZE"): e["write_over_known"] = round(v["WRITE_SIZE"] / known[k], 4)
    if k == "calib_gather16":
        e["known_bytes_requested"] = nq * 16 + nq * 4
        e["known_bytes_64B_lines"] = nq * 64 + nq * 4
        if v.get("FETCH_SIZE"):
            e["fetch_over_requested"] = round(v["FETCH_SIZE"] / (nq * 20), 4)
            e["fetch_over_lines"] = round(v["FETCH_SIZE"] / (nq * 68), 4)
    doc["patterns"][k] = e
json.dump(doc, open(os.path.join(out, "fetch_calibration.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
PY
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
