"""Host logic of the batched driver (N1) on CPU with a stand-in model: batching by image size, order
preservation through the double-buffered pipeline, KITTI 16-bit encoding."""
import numpy as np
import torch

from nmrf_amd.driver import StereoStream, batches, encode_kitti_disp


def _pairs(shapes):
    for i, (h, w) in enumerate(shapes):
        yield i, torch.full((3, h, w), float(i)), torch.full((3, h, w), float(i) + 0.5)


def test_batches_group_equal_sizes_in_order():
    shapes = [(4, 6)] * 5 + [(8, 6)] * 2 + [(4, 6)]
    groups = list(batches(_pairs(shapes), 3))
    assert [len(g) for g in groups] == [3, 2, 2, 1]
    assert [g[0][0] for g in groups] == [0, 3, 5, 7]


def test_stream_preserves_order_and_values():
    calls = []

    def model(sample):                         # "disparity" = mean of left + right, per image
        calls.append(sample["img1"].shape[0])
        return {"disp": (sample["img1"].mean(1) + sample["img2"].mean(1))}

    shapes = [(4, 6)] * 7 + [(2, 3)] * 2
    out = list(StereoStream(model, device="cpu", batch=4).run(_pairs(shapes)))
    assert [k for k, _ in out] == list(range(9))
    assert calls == [4, 3, 2]
    for k, d in out:
        assert d.shape == shapes[k] and torch.allclose(d, torch.full(shapes[k], 2.0 * k + 0.5))


def test_kitti_16bit_encoding():
    d = np.array([[0.0, 1.0, 10.5, 255.99, 300.0]])
    assert encode_kitti_disp(d).tolist() == [[0, 256, 2688, 65533, 65535]]
    assert encode_kitti_disp(d).dtype == np.uint16
