#!/usr/bin/env python3
"""HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh, corrected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE reports half of the bytes of a
wide (16 B per lane) coalesced streaming read -> doubled for the streaming kernels; WRITE_SIZE is calibrated in the
same run on k_synth, whose stores are known exactly (2 + 4 + 4 bytes per call at 100k x 10k = 12.0e9 B in the bench run).
    python tools/pmc_traffic.py <bench pmc csv> <configs pmc csv> <tag>"""
import csv
import json
import sys


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out.setdefault(r['kernel'], {})[r['counter']] = (float(r['mean_value']), int(r['dispatches']), float(r['mean_duration_ns']))
    return out


def pick(d, needle):
    for k in d:
        if needle in k:
            return k, d[k]
    return None, None


def main():
    bench, configs, tag = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {"round": tag, "source": "profiles/%s_bench_pmc_fetch_write.csv, profiles/%s_configs_pmc_fetch_write.csv "
                                   "(tools/profile_round.sh %s)" % (tag, tag, tag),
           "corrections": "FETCH_SIZE x 2 for the 16 B/lane streaming kernels (guide: gfx950 reports half); WRITE_SIZE x "
                          "the factor that makes k_synth's stores come out at their known byte count in the same run"}
    _, syn = pick(bench, 'k_synth(')
    wfac = 1.0
    if syn and 'WRITE_SIZE' in syn:
        wfac = 12.0e9 / (syn['WRITE_SIZE'][0] * 1024.0)
        res["write_calibration"] = {"k_synth_write_kib_raw": syn['WRITE_SIZE'][0], "known_bytes": 12.0e9, "factor": wfac}
    def entry(d, needle, algo):
        k, v = pick(d, needle)
        if not v or 'FETCH_SIZE' not in v:
            return None
        f = v['FETCH_SIZE'][0] * 1024.0 * 2.0
        w = v.get('WRITE_SIZE', (0.0, 0, 0.0))[0] * 1024.0 * wfac
        return {"kernel": k, "fetch_kib_raw": v['FETCH_SIZE'][0], "write_kib_raw": v.get('WRITE_SIZE', (0.0,))[0],
                "fetch_bytes": f, "write_bytes": w, "bytes_per_launch": f + w, "algorithmic_bytes_per_launch": algo,
                "ratio": (f + w) / algo if algo else None}
    res["k_call_filter"] = entry(bench, 'k_call_filter_v4<3, 1, true, false', 20.0e9) or entry(bench, 'k_call_filter_v2<3, true, false', 20.0e9)
    res["k_stream_probe"] = entry(bench, 'k_stream_probe<3, 2>', 20.0e9)
    res["k_cf_reduce"] = entry(bench, 'k_cf_reduce', None)
    res["k_locus_count"] = entry(bench, 'k_locus_count_v2<', 4.0e9)
    res["k_assoc_scan"] = entry(bench, 'k_assoc_scan_few<1, false>', 4.0e9) or entry(bench, 'k_assoc_scan<1, false>', 4.0e9)
    res["config1_k_locus_count"] = entry(configs, 'k_locus_count_v3<4', 4.0e7)
    res["config2_k_call_filter_gs_planar"] = entry(configs, 'k_call_filter_gs<true, true', 50000 * 5000 * 60.0)
    res["config2_k_call_filter_gs_interleaved"] = entry(configs, 'k_call_filter_gs<false, true', 50000 * 5000 * 72.0)
    res["config2_k_call_filter_fast"] = (entry(configs, 'k_call_filter_fast<12, true, 1>', 50000 * 5000 * 60.0) or
                                         entry(configs, 'k_call_filter_fast<12, true, true>', 50000 * 5000 * 60.0) or
                                         entry(configs, 'k_call_filter_fast<12, true>', 50000 * 5000 * 60.0))
    # the same filters on planes left interleaved (as cyvcf2 hands them): every column of a plane is fetched
    res["config2_interleaved_k_call_filter_fast"] = entry(configs, 'k_call_filter_fast<16, true, 2>', 50000 * 5000 * 72.0)
    if res["k_call_filter"]:
        res["k_call_filter_bytes_per_launch"] = res["k_call_filter"]["bytes_per_launch"]
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
