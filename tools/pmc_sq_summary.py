"""Per-kernel averages of an arbitrary rocprofv3 --pmc pass (SQ issue / wait breakdown).
usage: python tools/pmc_sq_summary.py <dir with *_counter_collection.csv> out.txt"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(dict))
    for path in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                name = re.sub(r'\(anonymous namespace\)::', '', row['Kernel_Name'])
                name = re.sub(r'^void ', '', re.sub(r'\(.*', '', name))[:60]
                d = acc[name][row['Counter_Name']]
                d[row['Dispatch_Id']] = d.get(row['Dispatch_Id'], 0.0) + float(row['Counter_Value'])
    rows = []
    for name, ctrs in acc.items():
        avg = {c: sum(v.values()) / len(v) for c, v in ctrs.items()}
        n = len(next(iter(ctrs.values())))
        rows.append((avg.get('SQ_WAVE_CYCLES', 0.0) * n, name, n, avg))
    rows.sort(reverse=True)
    ctr_names = sorted({c for r in rows for c in r[3]})
    lines = ['# rocprofv3 --pmc ' + ' '.join(ctr_names) + '  (per-launch averages, summed over XCC instances; SQ_WAVE/WAIT/ACTIVE are quad-cycles)',
             f'{"kernel":60s} {"launches":>8s} ' + ' '.join(f'{c[-22:]:>22s}' for c in ctr_names) + '   wait_any%  wait_inst%  active%  mfma_busy/wave_cyc']
    for _, name, n, avg in rows[:24]:
        wc = avg.get('SQ_WAVE_CYCLES', 0.0) or 1.0
        extra = '   %8.1f  %9.1f  %7.1f  %10.3f' % (100 * avg.get('SQ_WAIT_ANY', 0) / wc, 100 * avg.get('SQ_WAIT_INST_ANY', 0) / wc,
                                                      100 * avg.get('SQ_ACTIVE_INST_ANY', 0) / wc, avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / wc)
        lines.append(f'{name:60s} {n:8d} ' + ' '.join(f'{avg.get(c, 0.0):22.0f}' for c in ctr_names) + extra)
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(l[:260] for l in lines[:12]))


if __name__ == '__main__':
    main()
