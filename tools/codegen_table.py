"""Register / scratch budget of the gfx950 kernels of one csrc file, as the compiler allocates them (hipcc -S --offload-device-only, the
.amdhsa kernel descriptors' metadata): {kernel symbol: {sgpr, vgpr, scratch, vspill}}.

  python tools/codegen_table.py cat_amd/csrc/conv_pk.hip [--write tests/golden/codegen_conv_pk.json]

tests/test_codegen.py holds the LDS-tile kernels to the committed table: their code generation is fragile (round 5: two extra epilogue
branches with a feature switched off cost every tconv launch 5-8 %, SGPRs 101 -> 103), so a register-count change must be a decision that
comes with an A/B on hardware, not a side effect of an edit elsewhere in the file."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def table(src):
    src = src if os.path.isabs(src) else os.path.join(ROOT, src)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--offload-device-only', src, '-o', out], check=True,
                       capture_output=True)
        text = open(out).read()
    res = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', text, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(k + r':\s+(\d+)', body).group(1))
        res[m.group(1)] = dict(sgpr=g(r'\.sgpr_count'), vgpr=g(r'\.vgpr_count'), scratch=g(r'\.private_segment_fixed_size'),
                               vspill=g(r'\.vgpr_spill_count'))
    return res


if __name__ == '__main__':
    t = table(sys.argv[1])
    if '--write' in sys.argv:
        json.dump(t, open(sys.argv[sys.argv.index('--write') + 1], 'w'), indent=0, sort_keys=True)
    for k, v in sorted(t.items()):
        print('%-110s sgpr %3d vgpr %3d scratch %3d' % (k, v['sgpr'], v['vgpr'], v['scratch']))
