"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a small text table for profiles/.
usage: python tools/rocprof_summary.py gpurun_out/prof_r01/bench_results.db profiles/r01_kernel_stats.txt [steps]"""
import re
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    lines = ['# rocprofv3 --kernel-trace --stats summary (durations in us; whole process incl. model construction)',
             f'# per-step columns divide by {steps:g} profiled steps', '',
             f'{"kernel":78s} {"calls":>7s} {"total_us":>11s} {"avg_us":>9s} {"pct":>6s} {"us/step":>10s}']
    for name, calls, tot, avg, pct in rows[:60]:
        name = re.sub(r'\(anonymous namespace\)::', '', str(name))
        name = re.sub(r'\(.*', '', name)[:78]
        lines.append(f'{name:78s} {calls:7d} {tot:11.1f} {avg:9.2f} {pct:6.2f} {tot / steps:10.1f}')
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:25]))


if __name__ == '__main__':
    main()
