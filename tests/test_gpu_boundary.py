"""GPU tests of the drop-in boundary: the reference trainers' own call sequences through the ``pc_processor`` shim
(tasks/pmf/trainer.py:33-39,80-98,139-147; tasks/epmf/trainer.py:195-198) and the pieces they need."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import loader_ref  # noqa: E402
from tests.test_oracle_golden import LOADER_TRAIN_CASES, loader_train_oracle, _all_colours  # noqa: E402


class _Frames(object):
    """the duck type the loaders read from the reference's SemanticKitti (parser.py:7-227)"""

    def __init__(self, seed, npts, h, w):
        from PIL import Image
        M, self.pts, self.sem, self.img, lut = loader_ref.synthetic_frame(seed, npts, h, w)
        self.proj_matrix, self.class_map_lut = {"00": M}, lut
        self._image = Image.fromarray(self.img)

    def loadDataByIndex(self, i):
        return self.pts, self.sem, np.zeros_like(self.sem)

    def loadImage(self, i):
        return self._image

    def parsePathInfoByIndex(self, i):
        return "00", "000000"

    def __len__(self):
        return 1


def test_color_jitter_exact():
    """pmf_color_jitter against the Pillow-pinned oracle: every colour, every operation order class, interpolating and
    extrapolating blend factors, negative and positive hue shifts -- bit for bit"""
    from oracle import color_jitter_ref as CJ
    from pmf_amd.dataset.perspective_view_loader import ColorJitter
    img = _all_colours()
    cj = ColorJitter(0.4, 0.4, 0.4, 0.1)
    cases = [([0, 1, 2, 3], [0.61, 1.37, 0.72, -0.08]), ([3, 2, 1, 0], [1.399, 0.6, 1.2, 0.1]),
             ([2, 0, 3, 1], [1.0, None, 0.0, 0.031]), ([1, 3, 0, 2], [0.0, 1.0, 1.4, -0.1])]
    torch.manual_seed(11)
    cases.append(cj.draw())
    for order, fac in cases:
        got = cj.apply(torch.from_numpy(img).cuda(), order, fac).cpu().numpy()
        want = CJ.color_jitter(img, order, fac)
        assert np.array_equal(got, want), (order, fac, int((got != want).sum()))
    # a small odd-sized frame (grid tail, contrast mean over few pixels) and the draw order of torchvision.get_params
    rng = np.random.default_rng(3)
    small = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for seed in range(4):
        torch.manual_seed(seed)
        order, fac = cj.draw()
        torch.manual_seed(seed)
        assert (order, fac) == CJ.draw_params(CJ.jitter_ranges(0.4, 0.4, 0.4, 0.1))
        got = cj.apply(torch.from_numpy(small).cuda(), order, fac).cpu().numpy()
        assert np.array_equal(got, CJ.color_jitter(small, order, fac))
    assert ColorJitter(0, 0, 0, 0).ranges == (None, None, None, None)
    with pytest.raises(RuntimeError):
        cj.apply(torch.from_numpy(small), [0, 1, 2, 3], [1.0, 1.0, 1.0, 0.0])
    with pytest.raises(ValueError):
        ColorJitter(hue=0.7)


@pytest.mark.parametrize("case", LOADER_TRAIN_CASES, ids=lambda c: c[0])
def test_reference_trainer_loader_call(case, golden):
    """tasks/pmf/trainer.py:139-142 verbatim through the shim: PerspectiveViewLoader(dataset=trainset, config=...,
    is_train=True, pcd_aug=False, img_aug=True, use_padding=True); the item equals the reference-run fixture g13 (and
    the oracle) except where the inverse-rotated source coordinate sits on a nearest-neighbour rounding edge"""
    import pc_processor
    tag, seed, npts, h, w, ht, wt, hp, wp = case
    g = golden("g13_loader_train")
    cfg = {"augmentation": {"img_jitter": [0.4, 0.4, 0.4, 0.1]},
           "sensor": {"proj_h": h, "proj_w": w, "proj_ht": ht, "proj_wt": wt, "h_pad": hp, "w_pad": wp}}
    train_pv_loader = pc_processor.dataset.PerspectiveViewLoader(
        dataset=_Frames(seed, npts, h, w), config=cfg, is_train=True, pcd_aug=False, img_aug=True, use_padding=True)
    for rep in range(2):
        torch.manual_seed(100 * seed + rep)
        feat, mask, label = train_pv_loader[0]
        got = torch.cat([feat, mask[None], label[None]], 0).cpu().numpy()
        want = g["train.%s.%d" % (tag, rep)]
        assert got.shape == want.shape == (10, ht, wt)
        assert np.array_equal(want, loader_train_oracle(seed, rep, *case[2:]))
        bad = np.argwhere((got != want).any(0))
        assert bad.shape[0] <= 1e-4 * ht * wt + 2, (tag, rep, bad.shape[0])
    # validation loader of the same trainer (trainer.py:144-147): jitter off, CenterCrop + Pad
    val = pc_processor.dataset.PerspectiveViewLoader(dataset=_Frames(seed, npts, h, w), config=cfg, is_train=False,
                                                     use_padding=True)
    f, m, l = val[0]
    assert f.shape == (8, h, w) and val.img_jitter is None
