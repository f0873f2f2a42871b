"""CPU: the code generation of the LDS-tile convolution kernels (csrc/conv_pk.hip) is pinned -- hipcc cross-compiles gfx950 here, and the
SGPR / VGPR / scratch counts of every tconv_kernel / tstage1_kernel instantiation must equal the committed table
(tests/golden/codegen_conv_pk.json, written by tools/codegen_table.py).  Round-5 verdict, robustness #14: these kernels lost 5-8 % per launch
to two extra epilogue branches of a switched-off feature (SGPRs 101 -> 103); a changed count now fails the CPU suite and has to be re-recorded
on purpose, together with an A/B on hardware (profiles/r06_tconv_ab.txt shows the form such a record takes).
The wide DMA-staged tiles only have to keep their occupancy: two workgroups per CU = at most 256 registers, no scratch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_lds_tile_kernels_keep_their_register_budget():
    import codegen_table
    want = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'codegen_conv_pk.json')))
    got = codegen_table.table('cat_amd/csrc/conv_pk.hip')
    assert sorted(got) == sorted(want), 'kernel set of conv_pk.hip changed: re-record with tools/codegen_table.py --write (after an A/B on hardware)'
    diff = {k: (want[k], got[k]) for k in want if want[k] != got[k]}
    assert not diff, 'register / scratch counts changed (committed, now): %s' % diff
    # the budgets the launch rules rely on: the 8 x 16 tiles of up to 5 N tiles run two workgroups per CU with room to spare
    for k, v in got.items():
        if 'tconv_kernelILi' in k and 'ELi16E' in k:
            assert v['scratch'] == 0 and v['vgpr'] <= 256, (k, v)


def test_wide_dma_tiles_keep_two_workgroups_per_cu():
    import codegen_table
    for src, names in (('cat_amd/csrc/conv_ksum.hip', ('ksum_kernel3', 'ksum_kernel4')),
                       ('cat_amd/csrc/conv_igemm.hip', ('conv_fwd32d_kernel', 'conv_dgrad32d_kernel', 'conv_wgrad32d_kernel'))):
        t = codegen_table.table(src)
        hit = {k: v for k, v in t.items() if any(n in k for n in names)}
        assert len(hit) >= len(names), (src, sorted(t))
        for k, v in hit.items():
            assert v['vgpr'] <= 256 and v['scratch'] == 0 and v['vspill'] == 0, (k, v)
