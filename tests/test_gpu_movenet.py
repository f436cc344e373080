"""GPU parity: MoveNet crop / coordinate update / crop-region kernels and the MultiPose head decode vs
oracle/movenet.py through the C ABI (BASELINE configs[3]).  Bit-exact: int32 crops, float32
keypoints, float64 crop regions, float32 decode rows."""
import numpy as np
import pytest

from oracle import movenet as omv
from posehip import movenet as mv

pytestmark = pytest.mark.gpu


def _regions(rng, n):
    out = [omv.init_crop_region(360, 640), omv.init_crop_region(640, 360)]
    while len(out) < n:
        y0, x0 = rng.uniform(-0.3, 0.6, 2)
        side = rng.uniform(0.2, 1.2)
        out.append({'y_min': y0, 'x_min': x0, 'y_max': y0 + side, 'x_max': x0 + side * 0.8, 'height': side,
                    'width': side * 0.8})
    return out[:n]


def test_crop_and_resize_matches_oracle():
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (5, 90, 160, 3), dtype=np.uint8)
    regions = _regions(rng, 5)
    for size in (32, 1, 47):
        got = mv.crop_and_resize_batch(frames, regions, (size, size))
        assert got.dtype == np.int32 and got.shape == (5, size, size, 3)
        for b in range(5):
            np.testing.assert_array_equal(got[b], omv.crop_and_resize(frames[b], regions[b], (size, size)))
    one = mv.crop_and_resize(frames[0:1], regions[0], (16, 16))
    np.testing.assert_array_equal(one[0], omv.crop_and_resize(frames[0], regions[0], (16, 16)))
    with pytest.raises(ValueError):
        mv.crop_and_resize_batch(frames, regions, (16, 24))


def test_update_and_next_crop_region_bit_exact():
    rng = np.random.default_rng(11)
    B = 64
    k = rng.uniform(0, 1, (B, 1, 17, 3)).astype(np.float32)
    k[:, :, :, 2] = rng.choice([0.05, 0.19, 0.2, 0.21, 0.6, 0.95], (B, 1, 17)).astype(np.float32)
    k[3, 0, [5, 6, 11, 12], 2] = 0.9          # a certainly visible torso
    k[4, 0, [11, 12], 2] = 0.1                # hips hidden -> default crop
    regions = _regions(rng, B)
    for ih, iw in ((360, 640), (640, 360), (512, 512)):
        got_k, got_r = mv.update_and_determine_batch(k, regions, ih, iw)
        n_default = 0
        for b in range(B):
            want_k = omv.update_keypoints(k[b:b + 1], regions[b], ih, iw)
            np.testing.assert_array_equal(got_k[b:b + 1].view(np.uint32), want_k.view(np.uint32))
            want_r = omv.determine_crop_region(want_k, ih, iw)
            for key in want_r:
                assert np.float64(got_r[b][key]).view(np.uint64) == np.float64(want_r[key]).view(np.uint64), (b, key)
            n_default += want_r == omv.init_crop_region(ih, iw)
        assert 0 < n_default < B               # both branches taken
    r = mv.determine_crop_region(got_k[3:4], 512, 512)
    assert r == omv.determine_crop_region(got_k[3:4], 512, 512)


def test_run_inference_round_trip():
    rng = np.random.default_rng(5)
    image = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)
    region = omv.init_crop_region(120, 200)
    seen = {}

    def model(x):
        seen['x'] = x
        out = np.zeros((1, 1, 17, 3), np.float32)
        out[0, 0, :, 0], out[0, 0, :, 1], out[0, 0, :, 2] = 0.5, 0.25, 0.8
        return {'output_0': out}

    k = mv.run_inference(model, image, region, (48, 48))
    assert seen['x'].shape == (1, 48, 48, 3) and seen['x'].dtype == np.int32
    np.testing.assert_array_equal(seen['x'][0], omv.crop_and_resize(image, region, (48, 48)))
    np.testing.assert_array_equal(k, omv.update_keypoints(model(None)['output_0'], region, 120, 200))


def test_multipose_decode_matches_specification():
    rng = np.random.default_rng(21)
    # (64, 64, 6) = BASELINE configs[3]: batch 64 of the 256x256 MultiPose head (64x64 grid, 6 people)
    for B, G, P in ((3, 64, 6), (64, 64, 6), (2, 16, 4), (1, 9, 12)):
        center = rng.uniform(0, 1, (B, G, G)).astype(np.float32) ** 4
        center[0, 0, 0] = center[0, G - 1, G - 1] = 1.0        # maxima on the border / corners
        heat = rng.uniform(0, 1, (B, G, G, 17)).astype(np.float32)
        regr = rng.normal(0, G / 8, (B, G, G, 34)).astype(np.float32)
        offs = rng.uniform(-0.5, 0.5, (B, G, G, 34)).astype(np.float32)
        box = rng.uniform(1, G / 2, (B, G, G, 2)).astype(np.float32)
        if B > 1:
            center[1] = 0.0                                     # nobody: all rows zero
        got = mv.multipose_decode(center, heat, regr, offs, box, P, 0.3)
        assert got.shape == (B, P, 56)
        for b in range(B):
            want = omv.multipose_decode(center[b], heat[b], regr[b], offs[b], box[b], P, 0.3)
            np.testing.assert_array_equal(got[b].view(np.uint32), want.view(np.uint32))
        if B > 1:
            assert not got[1].any()
        assert got[0, 0, 55] == 1.0
    with pytest.raises(ValueError):
        mv.multipose_decode(center, heat[:, :, :, :16], regr, offs, box)
