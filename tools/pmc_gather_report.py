"""Turns the two PMC passes of tools/pmc_gather.py (FETCH_SIZE, WRITE_SIZE counter_collection CSVs)
into profiles/r01_gather_pmc.json: corrections from the calibration launch (known bytes), corrected
traffic of the bench-shaped launch.   usage: pmc_gather_report.py fetch.csv write.csv STRIDE out.json"""
import csv
import json
import sys

fetch_csv, write_csv, stride, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
ROW = int(sys.argv[5]) if len(sys.argv) > 5 else 200            # row bytes
N = int(sys.argv[6]) if len(sys.argv) > 6 else 770_000          # rows of the measured launch
TABLE = int(sys.argv[7]) if len(sys.argv) > 7 else 2_449_029


def gather_rows(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if "k_gather_rows" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    assert len(rows) == 2, f"{path}: expected the calibration and the measured launch, got {len(rows)}"
    return float(rows[0]["Counter_Value"]), float(rows[1]["Counter_Value"]), rows[1]["Kernel_Name"].split("(")[0]


N_CAL = 4_000_000
f_cal, f_meas, name = gather_rows(fetch_csv, "FETCH_SIZE")
w_cal, w_meas, _ = gather_rows(write_csv, "WRITE_SIZE")
cal_read = N_CAL * (ROW + 4)          # rows + int32 index, every byte once
cal_write = N_CAL * ROW
fc = cal_read / (f_cal * 1024)
wc = cal_write / (w_cal * 1024)
read = f_meas * 1024 * fc
write = w_meas * 1024 * wc
alg = 2 * ROW + 8
doc = {
    "kernel": name.replace("void ", ""),
    "tool": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/pmc_gather.py + tools/pmc_gather_report.py",
    "shape": {"rows": N, "row_bytes": ROW, "src_stride_bytes": stride, "table_rows": TABLE, "index": "int32 random"},
    "calibration": {"rows": N_CAL, "index": "arange over a dense table (every byte once, 800 MB >> Infinity Cache)",
                    "FETCH_SIZE_KB": f_cal, "WRITE_SIZE_KB": w_cal, "fetch_correction": fc, "write_correction": wc,
                    "note": "FETCH_SIZE reads 1/2 of the fetched bytes on gfx950 (MI355X_MICROARCH.md, HBM); "
                            "WRITE_SIZE reads a few % high for 8-B non-temporal stores"},
    "measured": {"FETCH_SIZE_KB": f_meas, "WRITE_SIZE_KB": w_meas},
    "corrected_bytes": {"read": read, "write": write, "total": read + write},
    "traffic_bytes_per_row": (read + write) / N,
    "algorithmic_bytes_per_row": alg,
    "read_amplification": read / (N * (ROW + 4)),
    "comment": ("rows start on a 128-B fetch granule (resident table padded to 256 B per 200-B row): "
                "2 granules per row instead of 2.56 for the dense layout (which measured 520 B/row)")
               if (stride == 256 and ROW == 200) else
               ("256-B rows are exactly two fetch granules" if ROW == 256 else "dense rows"),
}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc["corrected_bytes"]), doc["traffic_bytes_per_row"], doc["read_amplification"])
