"""A16 (parity unpinned): the CPU restatement of the superpixel-guided downsample on hand-made cells."""
import numpy as np

from oracle import superpixel_oracle as SO


def test_modes_are_segment_means_ordered_by_size():
    disp = np.zeros((1, 8, 16), np.float32)
    lab = np.zeros((1, 8, 16), np.int32)
    # cell 0: segment 7 covers 40 pixels at disparity 10, segment 3 covers 24 pixels at 20 (4 of them invalid)
    lab[0, :, :8] = 7
    disp[0, :, :8] = 10
    lab[0, 5:, :8] = 3
    disp[0, 5:, :8] = 20
    disp[0, 7, :4] = 0
    # cell 1: one segment, two values
    lab[0, :, 8:] = 1
    disp[0, :, 8:] = 4
    disp[0, 0, 8:] = 8
    out = SO.downsample_disp(disp, lab, 3)
    assert out.shape == (1, 1, 2, 3)
    assert np.allclose(out[0, 0, 0], [10, 20, 0])
    assert np.allclose(out[0, 0, 1], [(56 * 4 + 8 * 8) / 64, 0, 0])


def test_ties_by_label_and_truncation_to_k():
    disp = np.ones((1, 8, 8), np.float32)
    lab = np.repeat(np.arange(4, dtype=np.int32)[::-1], 16).reshape(1, 8, 8)     # four segments of 16 pixels: labels 3,2,1,0
    disp[0] = (lab[0] + 1) * 5
    out = SO.downsample_disp(disp, lab, 2)
    assert np.allclose(out[0, 0, 0], [5, 10])                                      # equal counts -> smaller label first
