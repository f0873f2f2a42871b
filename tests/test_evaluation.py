"""CPU: the host side of evaluate_model (cat_amd/distillers/evaluation.py) -- image conversion pinned to the reference's utils/util.py
by tests/golden/eval_utils.npz (tools/make_golden.py, GOLDEN_ONLY=eval), and the bookkeeping of reference
inception_distiller.py:204-281 / spade_distiller.py:96-180 driven with stub networks (the generator passes themselves are GPU code)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

import helpers as H
from cat_amd.distillers import evaluation as E
from cat_amd.distillers.base_inception_distiller import BaseInceptionDistiller
from cat_amd.distillers.base_spade_distiller import BaseSPADEDistiller


def test_image_conversion_matches_reference_utils():
    g = H.load('eval_utils.npz')
    img, lab = torch.from_numpy(g['img']), torch.from_numpy(g['lab'])
    assert np.array_equal(E.tensor2im(img), g['img_u8'])
    assert np.array_equal(E.tensor2im(img[:1]), g['gray_u8'])
    assert np.array_equal(E.label_colormap(37), g['cmap37']) and np.array_equal(E.label_colormap(20), g['cmap20'])
    assert np.array_equal(E.tensor2label(lab, 37), g['lab_u8'])


class _Student(torch.nn.Module):
    pass


def _inception_stub(tmp_path, mode='aligned', dataroot='database/maps', direction='AtoB'):
    m = BaseInceptionDistiller.__new__(BaseInceptionDistiller)
    m.opt = Namespace(dataset_mode=mode, log_dir=str(tmp_path), dataroot=dataroot, direction=direction)
    m.netG_student = _Student()
    m.best_fid, m.best_mIoU, m.fids, m.mIoUs, m.is_best = 1e9, -1e9, [], [], False
    calls = []

    def feed(batch):
        calls.append(('in', batch['A_paths'][0]))
        m.real_A, m.real_B, m.image_paths = batch['A'], batch.get('B'), batch['A_paths']
    m.set_input = feed
    m.set_single_input = feed

    def test():
        assert not m.netG_student.training          # inference runs with eval-mode norms
        m.Sfake_B, m.Tfake_B = m.real_A * 0.5, -m.real_A
    m.test = test
    m.eval_dataloader = [{'A': torch.full((2, 3, 4, 4), 0.1 * (i + 1)), 'B': torch.zeros(2, 3, 4, 4), 'A_paths': [f'/x/im{i}a.jpg', f'/x/im{i}b.jpg']}
                         for i in range(7)]
    return m, calls


def test_evaluate_model_bookkeeping(tmp_path):
    m, calls = _inception_stub(tmp_path)
    with pytest.raises(RuntimeError, match='fid_fn'):
        m.evaluate_model(1)
    seen = {}

    def fid(fakes):
        seen['fakes'] = fakes
        return seen.setdefault('next', 30.0)
    m.fid_fn = fid
    r = m.evaluate_model(5)
    assert r == {'metric/fid': 30.0, 'metric/fid-mean': 30.0, 'metric/fid-best': 30.0} and m.is_best and m.best_fid == 30.0
    assert m.netG_student.training                   # back in train mode
    assert len(seen['fakes']) == 7 and seen['fakes'][0].shape == (2, 3, 4, 4) and seen['fakes'][0].is_contiguous()
    d = os.path.join(str(tmp_path), 'eval', '5')
    assert sorted(os.listdir(d)) == ['Sfake', 'Tfake', 'input', 'real']
    assert len(os.listdir(os.path.join(d, 'Sfake'))) == 10 and 'im0a.png' in os.listdir(os.path.join(d, 'input'))   # first 10 samples only
    for v, best, mean, is_best in ((40.0, 30.0, 35.0, False), (20.0, 20.0, 30.0, True), (50.0, 20.0, (40 + 20 + 50) / 3, False)):
        seen['next'] = v
        r = m.evaluate_model(6)
        assert r['metric/fid'] == v and r['metric/fid-best'] == best and abs(r['metric/fid-mean'] - mean) < 1e-9 and m.is_best == is_best
    assert len(m.fids) == 3
    r = m.evaluate_model(7, save_image=True)
    assert len(os.listdir(os.path.join(str(tmp_path), 'eval', '7', 'Tfake'))) == 14


def test_evaluate_model_unaligned_and_miou(tmp_path):
    m, calls = _inception_stub(tmp_path, mode='unaligned', dataroot='database/cityscapes', direction='BtoA')
    m.fid_fn = lambda fakes: 10.0
    with pytest.raises(RuntimeError, match='miou_fn'):
        m.evaluate_model(1)
    names_seen = []
    m.miou_fn = lambda fakes, names: names_seen.extend(names) or 0.4
    r = m.evaluate_model(2)
    assert r['metric/mIoU'] == 0.4 and r['metric/mIoU-best'] == 0.4 and m.is_best and names_seen[:2] == ['im0a', 'im0b']
    assert sorted(os.listdir(os.path.join(str(tmp_path), 'eval', '2'))) == ['Sfake', 'Tfake', 'input']      # no 'real' for unaligned data
    m.eval_dataloader = None
    with pytest.raises(RuntimeError, match='eval_dataloader'):
        m.evaluate_model(3)


def test_spade_evaluate_model(tmp_path):
    m = BaseSPADEDistiller.__new__(BaseSPADEDistiller)
    m.opt = Namespace(log_dir=str(tmp_path), dataroot='database/cityscapes-origin', no_fid=True, no_mIoU=False, input_nc=35)
    student = _Student()
    m.modules_on_one_gpu = Namespace(netG_student=student)
    m.best_fid, m.best_mIoU, m.fids, m.mIoUs, m.is_best = 1e9, -1e9, [], [], False

    def feed(batch):
        m.input_semantics, m.real_B, m.image_paths = batch['sem'], batch['image'], batch['path']
    m.set_input = feed

    def test():
        assert not student.training
        m.Sfake_B, m.Tfake_B = m.real_B * 0.5, -m.real_B
    m.test = test
    sem = torch.zeros(1, 36, 4, 6)
    sem[:, 3] = 1.0
    m.eval_dataloader = [{'sem': sem, 'image': torch.zeros(1, 3, 4, 6), 'path': ['/c/frankfurt_000.png']}]
    m.miou_fn = lambda fakes, names: 0.25
    r = m.evaluate_model(9)
    assert r == {'metric/mIoU': 0.25, 'metric/mIoU-mean': 0.25, 'metric/mIoU-best': 0.25} and student.training
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(str(tmp_path), 'eval', '9', 'input', 'frankfurt_000.png')))
    assert im.shape == (4, 6, 3) and tuple(im[0, 0]) == tuple(E.label_colormap(37)[3])
