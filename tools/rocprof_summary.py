"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a small text table for profiles/.
usage: python tools/rocprof_summary.py gpurun_out/prof_r01/bench_results.db profiles/r01_kernel_stats.txt [steps] [--library]
--library: only the kernels of libcat_hip.so (drops the ATen fills / random fills / copies of model construction and warm-up that the
           whole-process trace contains), percentages re-based on what is left"""
import re
import sqlite3
import sys


def main():
    argv = [a for a in sys.argv[1:] if a != '--library']
    library = '--library' in sys.argv
    db, out = argv[0], argv[1]
    steps = float(argv[2]) if len(argv) > 2 else 1.0
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    if library:
        rows = [r for r in rows if not re.search(r'at::native|__amd_rocclr|at_cuda_detail|rocprim|hipcub', str(r[0]))]
        tot = sum(r[2] for r in rows) or 1.0
        rows = [(r[0], r[1], r[2], r[3], 100.0 * r[2] / tot) for r in rows]
    lines = ['# rocprofv3 --kernel-trace --stats summary (durations in us; ' + ('libcat_hip.so kernels only: ATen / runtime kernels of model '
             'construction filtered out)' if library else 'whole process incl. model construction)'),
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
