"""Which gfx950 kernels of a .hip source changed between two git revisions?  Compiles both versions to assembly and compares the
instruction streams per kernel symbol (labels normalised).  Used to show that a refactor left the validated default kernels
bit-identical when there is no GPU time to re-run the suite.

  python tools/isa_diff.py cat_amd/csrc/norm.hip [old_rev=HEAD~1] [new_rev=WORKTREE]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def source_at(path, rev):
    if rev == 'WORKTREE':
        return open(os.path.join(ROOT, path)).read()
    return subprocess.run(['git', '-C', ROOT, 'show', f'{rev}:{path}'], check=True, capture_output=True, text=True).stdout


def kernels(src, path, tmp, tag):
    inc = os.path.join(ROOT, os.path.dirname(path))
    f = os.path.join(tmp, tag + '.hip')
    open(f, 'w').write(src)
    out = os.path.join(tmp, tag + '.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', inc, '-S', '--offload-device-only', f, '-o', out], check=True,
                   capture_output=True)
    res, cur = {}, None
    for line in open(out):
        m = re.match(r'^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$', line)
        if m and not line.startswith('.'):
            cur = m.group(1)
            res[cur] = []
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith('s_endpgm'):
            cur = None
            continue
        if not t or t.startswith((';', '.')):
            continue
        res[cur].append(re.sub(r'\.LBB\d+_\d+', '.L', re.sub(r';.*', '', t)).strip())
    return {k: v for k, v in res.items() if v and not k.startswith('__hip_cuid')}


def main():
    path = sys.argv[1]
    old_rev = sys.argv[2] if len(sys.argv) > 2 else 'HEAD~1'
    new_rev = sys.argv[3] if len(sys.argv) > 3 else 'WORKTREE'
    with tempfile.TemporaryDirectory() as tmp:
        old = kernels(source_at(path, old_rev), path, tmp, 'old')
        new = kernels(source_at(path, new_rev), path, tmp, 'new')
    same = changed = 0
    new_by_body = {}
    for k, v in new.items():
        new_by_body.setdefault(tuple(v), []).append(k)
    for k, v in sorted(old.items()):
        if k in new:
            ok = new[k] == v
            print(('identical ' if ok else 'CHANGED   ') + k + ('' if ok else f'  ({len(v)} -> {len(new[k])} instructions)'))
        else:
            twins = new_by_body.get(tuple(v))
            ok = bool(twins)
            print(('identical ' if ok else 'MISSING   ') + k + (f'  (now {twins[0]})' if ok else ''))
        same += ok
        changed += not ok
    for k in sorted(set(new) - set(old)):
        print('new       ' + k)
    print(f'{same} identical, {changed} changed / missing, {len(set(new) - set(old))} new symbols')
    return 1 if changed else 0


if __name__ == '__main__':
    sys.exit(main())
