"""Fold rocprofv3 --pmc counter CSVs into per-kernel HBM bytes per launch, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes: FETCH_SIZE and WRITE_SIZE come from SEPARATE passes (they do not fit one), both are in KB, and on gfx950 FETCH_SIZE
counts 64 B per 128-B request for wide coalesced reads -> doubled.

  python tools/pmc_summary.py <dir with *_counter_collection.csv of both passes> profiles/<name>.json [min_launches]"""
import csv
import glob
import json
import os
import re
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    min_launches = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    acc = {}
    for path in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                name = re.sub(r'\(anonymous namespace\)::', '', row['Kernel_Name'])
                name = re.sub(r'^void ', '', re.sub(r'\(.*', '', name))
                ctr = row['Counter_Name']
                if ctr not in ('FETCH_SIZE', 'WRITE_SIZE'):
                    continue
                d = acc.setdefault(name, {}).setdefault(ctr, {})
                key = row.get('Dispatch_Id') or row.get('Correlation_Id')
                d[key] = d.get(key, 0.0) + float(row['Counter_Value'])     # one row per XCC/instance: sum them per dispatch
    table = {}
    for name, ctrs in acc.items():
        if not all(c in ctrs and len(ctrs[c]) >= min_launches for c in ('FETCH_SIZE', 'WRITE_SIZE')):
            continue
        e = {}
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            vals = list(ctrs[c].values())
            e[c] = {'launches': len(vals), 'avg_kb': round(sum(vals) / len(vals), 1)}
        e['hbm_bytes_per_launch'] = int(1024 * (2 * e['FETCH_SIZE']['avg_kb'] + e['WRITE_SIZE']['avg_kb']))
        table[name] = e
    table = dict(sorted(table.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['FETCH_SIZE']['launches']))
    json.dump(table, open(out, 'w'), indent=1)
    for k, v in list(table.items())[:12]:
        print(f"{k[:70]:70s} launches {v['FETCH_SIZE']['launches']:5d}  HBM MB/launch {v['hbm_bytes_per_launch'] / 1e6:9.1f}")


if __name__ == '__main__':
    main()
