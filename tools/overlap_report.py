"""Per model kernel: duration on a resident batch vs beside the data path, and where the data path's kernels
land relative to the model's phases.
usage: overlap_report.py <resident kernel_trace.csv> <with-data kernel_trace.csv> [last n steps = 32]
Steps are delimited by the fused Adam launch.  Model kernels = everything that is not a data-path kernel of
libspp_hip (sampling chain, delivery)."""
import collections
import csv
import sys

DATA = ("k_deliver", "k_seed_init", "k_hop_", "k_hop0", "k_bucket_", "k_rng_", "k_gpart_", "k_pack_remote", "k_serve_rows",
        "k_export", "k_gather_rows", "k_chain_")


def short(name):
    n = name.replace("void ", "")
    if n.startswith("Cijk_"):
        mt = n.split("_MT")[1].split("_")[0] if "_MT" in n else "?"
        return "GEMM " + n.split("_S_B")[0].replace("Cijk_", "") + " MT" + mt
    n = n.split("(")[0]
    for p in ("spp::", "at::native::", "(anonymous namespace)::", "at::cuda::", "rocprim::ROCPRIM_400200_NS::detail::"):
        n = n.replace(p, "")
    return n[:60] or name[:60]


def is_data(name):
    if "spp::" not in name:
        return False
    base = name.split("spp::", 1)[1]
    return base.startswith(DATA)


def load(path, nsteps):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append(dict(s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]), name=r["Kernel_Name"], q=r["Queue_Id"],
                         vgpr=r["VGPR_Count"], agpr=r["Accum_VGPR_Count"], lds=r["LDS_Block_Size"],
                         wg=int(r["Workgroup_Size_X"]), grid=int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) *
                         int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))
    rows.sort(key=lambda x: x["s"])
    adam = [x for x in rows if "FusedOptimizerTensorListMetadata" in x["name"]]
    if len(adam) < nsteps + 1:
        raise SystemExit(f"{path}: only {len(adam)} optimiser steps in the trace")
    lo, hi = adam[-nsteps - 1]["e"], adam[-1]["e"]
    win = [x for x in rows if x["s"] >= lo and x["e"] <= hi]
    return win, lo, hi


def overlap(a, ivs):
    """ns of interval a = (s, e) covered by the union of the sorted, possibly overlapping intervals ivs"""
    tot = 0
    cur_s = cur_e = None
    for s, e in ivs:
        if e <= a[0] or s >= a[1]:
            continue
        s, e = max(s, a[0]), min(e, a[1])
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    res, rlo, rhi = load(sys.argv[1], nsteps)
    dat, dlo, dhi = load(sys.argv[2], nsteps)
    print(f"steps compared: the last {nsteps} of each leg; step = {1e-3 * (rhi - rlo) / nsteps:.1f} us resident, "
          f"{1e-3 * (dhi - dlo) / nsteps:.1f} us beside the data path (under the profiler)")

    def model_table(win):
        t = collections.OrderedDict()
        for x in win:
            if is_data(x["name"]):
                continue
            k = (short(x["name"]), x["grid"])
            v = t.setdefault(k, dict(n=0, ns=0, vgpr=x["vgpr"], agpr=x["agpr"], lds=x["lds"], wg=x["wg"], ov_del=0, ov_chain=0))
            v["n"] += 1
            v["ns"] += x["e"] - x["s"]
        return t
    rt, dt_ = model_table(res), model_table(dat)
    deliver = sorted((x["s"], x["e"]) for x in dat if is_data(x["name"]) and "k_deliver" in x["name"])
    chain = sorted((x["s"], x["e"]) for x in dat if is_data(x["name"]) and "k_deliver" not in x["name"])
    for x in dat:
        if is_data(x["name"]):
            continue
        v = dt_[(short(x["name"]), x["grid"])]
        v["ov_del"] += overlap((x["s"], x["e"]), deliver)
        v["ov_chain"] += overlap((x["s"], x["e"]), chain)
    print(f"\n{'model kernel':62s} {'grid':>7s} {'wg':>5s} {'vgpr':>5s} {'agpr':>5s} {'lds':>6s} {'n/step':>6s} "
          f"{'alone us':>9s} {'beside us':>9s} {'x':>5s} {'+us/step':>8s} {'%t w/ deliver':>13s} {'%t w/ chain':>11s}")
    tot_a = tot_b = 0.0
    keys = sorted(set(rt) | set(dt_), key=lambda k: -(dt_.get(k, rt.get(k))["ns"]))
    for k in keys:
        a, b = rt.get(k), dt_.get(k)
        ref = b or a
        na = a["n"] / nsteps if a else 0
        ua = a["ns"] / a["n"] / 1e3 if a else 0.0
        ub = b["ns"] / b["n"] / 1e3 if b else 0.0
        pa = a["ns"] / nsteps / 1e3 if a else 0.0
        pb = b["ns"] / nsteps / 1e3 if b else 0.0
        tot_a += pa
        tot_b += pb
        if max(pa, pb) < 2.0:
            continue
        od = 100.0 * b["ov_del"] / b["ns"] if b else 0.0
        oc = 100.0 * b["ov_chain"] / b["ns"] if b else 0.0
        print(f"{k[0]:62s} {k[1]:7d} {ref['wg']:5d} {ref['vgpr']:>5s} {ref['agpr']:>5s} {ref['lds']:>6s} {na:6.1f} "
              f"{ua:9.1f} {ub:9.1f} {ub / ua if ua else 0:5.2f} {pb - pa:8.1f} {od:13.0f} {oc:11.0f}")
    print(f"{'sum of model kernel time per step':62s} {'':47s} {tot_a:9.1f} {tot_b:9.1f} {'':5s} {tot_b - tot_a:8.1f}")

    # the data path's kernels in the with-data leg
    print(f"\n{'data-path kernel (with-data leg)':40s} {'n/step':>7s} {'avg us':>8s} {'us/step':>8s} {'% of its time beside a GEMM':>28s} "
          f"{'beside other model kernels':>27s} {'model stream idle':>18s}")
    gemm = sorted((x["s"], x["e"]) for x in dat if not is_data(x["name"]) and x["name"].startswith("Cijk_"))
    other = sorted((x["s"], x["e"]) for x in dat if not is_data(x["name"]) and not x["name"].startswith("Cijk_"))
    dk = collections.OrderedDict()
    for x in dat:
        if not is_data(x["name"]):
            continue
        v = dk.setdefault(short(x["name"]), dict(n=0, ns=0, g=0, o=0))
        v["n"] += 1
        v["ns"] += x["e"] - x["s"]
        g = overlap((x["s"], x["e"]), gemm)
        anym = overlap((x["s"], x["e"]), sorted(gemm + other))
        v["g"] += g
        v["o"] += anym - g
    dsum = 0.0
    for k, v in sorted(dk.items(), key=lambda kv: -kv[1]["ns"]):
        dsum += v["ns"] / nsteps / 1e3
        print(f"{k:40s} {v['n'] / nsteps:7.2f} {v['ns'] / v['n'] / 1e3:8.1f} {v['ns'] / nsteps / 1e3:8.1f} {100.0 * v['g'] / v['ns']:28.0f} "
              f"{100.0 * v['o'] / v['ns']:27.0f} {100.0 * (v['ns'] - v['g'] - v['o']) / v['ns']:18.0f}")
    print(f"{'sum of data-path kernel time per step':40s} {'':7s} {'':8s} {dsum:8.1f}")

    # hardware queues (streams that share one are served in order, whatever the streams say)
    qs = collections.defaultdict(collections.Counter)
    for x in dat:
        kind = "model" if not is_data(x["name"]) else ("delivery" if "k_deliver" in x["name"] else "chain")
        qs[x["q"]][kind] += 1
    print("\nhardware queue -> dispatches in the compared steps of the with-data leg: " +
          "; ".join(f"queue {q}: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())) for q, c in sorted(qs.items())))

    # the model stream's own picture: busy / idle per step in both legs
    for tag, win, lo, hi in (("resident", res, rlo, rhi), ("beside the data path", dat, dlo, dhi)):
        m = sorted((x["s"], x["e"]) for x in win if not is_data(x["name"]))
        busy = overlap((lo, hi), m)
        g = sorted((x["s"], x["e"]) for x in win if not is_data(x["name"]) and x["name"].startswith("Cijk_"))
        print(f"\n{tag}: model kernels cover {busy / nsteps / 1e3:.1f} us of the {(hi - lo) / nsteps / 1e3:.1f} us step "
              f"(GEMMs {overlap((lo, hi), g) / nsteps / 1e3:.1f} us); no model kernel running {(hi - lo - busy) / nsteps / 1e3:.1f} us per step")


main()
