"""Summarise rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
/opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes) into per-kernel HBM traffic per launch.

Units/corrections applied (same guide): the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide (16 B/lane) coalesced streaming read -- every read of these kernels is a 16-byte LDS-DMA
or 16-byte load -- so the read side is doubled; WRITE_SIZE is used as reported (it matches the algorithmic
output bytes of the store-only stem kernel to 5 digits, which calibrates it for these access patterns).

usage: python tools/pmc_summary.py <fetch_csv> <write_csv> <out_json> [batch]
"""
import collections, csv, json, re, sys


def friendly(mangled):
    m = re.match(r"_ZN3amx23conv3d_k3_zmarch_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E(?:Lb(\d)E)?(?:Lb(\d)E)?(?:Lb(\d)E)?", mangled)
    if m:   # <T, NCK, QT, TY, TX, R, OUTMODE, NS, POOL, SPLIT, STEM>; same text as the launcher prints
        t, nck, qt, ty, tx, r, o, ns, pool, split, stem = m.groups()
        if stem == "1":     # the stem-fed 16 -> 16 launch (7 stem waves + loader, input ring of 16 planes: ZmStemCfg)
            return f"conv3d_k3_zmarch<{'f16' if t == 'DF16_' else 'bf16'},stem1->16->16,2x{ty}x{tx},c8+st7+ld1,r{r}/16{',pool' if pool == '1' else ''}>"
        return (f"conv3d_k3_zmarch<{'f16' if t == 'DF16_' else 'bf16'},{16*int(nck)}->{16*int(qt)},2x{ty}x{tx},"
                f"c8+l{2*int(nck)}+s{ns},r{r},o{o}{',pool' if pool == '1' else ''}>")
    m = re.match(r"_ZN3amx21conv3d_upcat16_kernelI(DF16_|DF16b)Li(\d+)E", mangled)
    if m:
        return f"conv3d_upcat16<{'f16' if m.group(1) == 'DF16_' else 'bf16'},2x8x32,c8+l3,r10/6,o{m.group(2)}>"
    m = re.match(r"_ZN3amx18conv3d_stem_kernelI(DF16_|DF16b)Li(\d+)E", mangled)
    if m:
        split = re.search(r"Lb1EEEv", mangled) is not None      # <T, Q, TY, TX, TZ, NC, R, SPLIT>
        return f"conv3d_stem<{'f16' if m.group(1) == 'DF16_' else 'bf16'}{'x2' if split else ''},q{m.group(2)},2x8x32,c8+l1,r10>"
    m = re.match(r"_ZN3amx19conv3d_k3_v2_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E"
                 r"(?:Li(\d+)ELi(\d+)ELb(\d)E(?:Lb(\d)E)?)?", mangled)
    if m:   # <T, WZ, WY, WX, NWZ, NWY, Q, NCH, OUTMODE, NLW, NBUF, SPLIT, MX>; same text as the launcher prints
        t, wz, wy, wx, nwz, nwy, q, nch, o, nlw, nbuf, split, mx = m.groups()
        name = "f16" if t == "DF16_" else "bf16"
        if mx == "1":
            name += "x2mx"
        elif split == "1":
            name += "x2"
        waves = f"w{int(nwz)*int(nwy)}" + (f"+l{nlw},b{nbuf}" if nlw and int(nlw) > 0 else "")
        return f"conv3d_k3_v2<{name},{int(wz)*int(nwz)}x{int(wy)*int(nwy)}x{wx},{waves},q{q},nch{nch},o{o}>"
    m = re.match(r"_ZN3amx19conv3d_k3_ks_kernelI(DF16_|DF16b)NS_5KsCfgILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEELb(\d)E", mangled)
    if m:   # <T, KsCfg<TZ, TY, TX, Q, QP, CPW, KW, CW, TEAMS, NH, NBUF>, PART>; same text as the launcher prints
        t, tz, ty, tx, q, qp, cpw, kw, cw, teams, nh, nbuf, part = m.groups()
        return (f"conv3d_k3_ks<{'f16' if t == 'DF16_' else 'bf16'},{tz}x{ty}x{tx},q{q},k{kw}x{cpw},c{cw},t{teams},h{nh},b{nbuf}"
                f"{',part' if part == '1' else ''}>")
    if "splitk_reduce_kernel" in mangled:
        return "splitk_reduce"
    if "conv3d_k3_zx_kernel" in mangled:
        m = re.search(r"ZxCfgT(?:ILi|<)(\d+)(?:ELi|, )(\d+)", mangled)
        return f"conv3d_k3_zx<f16x2mx,32->32,2x{m.group(1)}x{m.group(2)},m4+x4+cv4,r6>" if m else "conv3d_k3_zx<f16x2mx,32->32,m4+x4+cv4,r6>"
    for k in ("in_apply_pool_kernel", "in_apply_fast_kernel", "in_apply_kernel", "upsample2_trilinear_kernel", "in_finalize_kernel",
              "in_prereduce_kernel", "in_stats_kernel", "conv3d_stem2_kernel"):
        if k in mangled:
            return k
    m = re.match(r"_ZN3amx21conv3d_upmerge_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)E", mangled)
    if m:   # <T, Q, TZ, TY, NBUF, KS, SPLIT, LXT>
        t, q, tz, ty, nbuf, ks, split, lxt = m.groups()
        name = ("f16" if t == "DF16_" else "bf16") + ("x2" if split == "1" else "")
        return f"conv3d_upmerge<{name},q{q},{tz}x{int(ty) * 16 // int(lxt)}x{lxt},b{nbuf},k{32 * int(ks)}>"
    if "pool2_kernel" in mangled:
        return "pool2<max>" if "Li0EEE" in mangled else "pool2<avg>"
    # ViT engine kernels: keep the (demangled or mangled) name up to the argument list
    m = re.search(r"(attn_fwd_kernel|attn_prep_kernel|wsgemm_kernel|gemm_kernel|tokconv_kernel|tokstem_kernel|ln_rows_kernel)(<[^>]*>|I[A-Za-z0-9_]*?E(?=Ev|v))?", mangled)
    if m:
        return m.group(0)
    # rocprofv3 mis-demangles the bf16 instantiations ("<bool _Accum, int, E, 2, 32, ...>"): keep the name and the numbers
    m = re.search(r"amx::(\w+_kernel)<bool _Accum, (?:int|bool), E[L]?,? ?([^>]*)>", mangled)
    if m:
        return f"{m.group(1)}<bf16,{m.group(2).replace(' ', '')}>"
    return None


def collect(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = friendly(r["Kernel_Name"])
        if k:
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] * 1024.0 for k, v in agg.items()}, {k: v[0] for k, v in agg.items()}


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    fetch, nf = collect(fetch_csv, "FETCH_SIZE")
    write, nw = collect(write_csv, "WRITE_SIZE")
    res = {"_meta": {"batch_per_gpu": batch, "unit": "bytes per launch",
                     "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950 wide-read under-count), write = WRITE_SIZE x 1024",
                     "sources": [fetch_csv.split('gpurun_out/')[-1], write_csv.split('gpurun_out/')[-1]]}}
    for k in sorted(set(fetch) | set(write)):
        rd, wr = 2.0 * fetch.get(k, 0.0), write.get(k, 0.0)
        res[k] = {"fetch_size_raw": fetch.get(k, 0.0), "read_corrected": rd, "write": wr, "traffic": rd + wr,
                  "launches_sampled": [nf.get(k, 0), nw.get(k, 0)]}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k != "_meta":
            print(f"{k:56s} read {v['read_corrected']/1e6:9.1f} MB  write {v['write']/1e6:9.1f} MB")


if __name__ == "__main__":
    main()
