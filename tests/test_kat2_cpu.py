"""KAT-2: the known answers of the reference's own projection demo (model_util.py:449-481) for the oracle's
restatement of `tf.unique` + `unsorted_segment_min` + `scatter_nd` (oracle/ops_np.scatter_min_range): five points with
ranges (9, 7, 7, 9, 8) in cells (1,1) (1,1) (1,1) (1,3) (4,4) of a 5x5 grid, values all ones -> the two range-7 points
of cell (1,1) are BOTH kept and summed (2), the range-9 point there loses, (1,3) and (4,4) hold one point each;
min_r gathered back per point = (7, 7, 7, 9, 8)."""
import numpy as np

from oracle import ops_np as O


def test_kat2_scatter_of_ones():
    r = np.array([9.0, 7.0, 7.0, 9.0, 8.0], np.float32)
    iRow, iCol = np.array([1, 1, 1, 1, 4]), np.array([1, 1, 1, 3, 4])
    # the demo keys tf.unique with iRow * 1800 + iCol (:460); any injective key of (row, col) gives the same segments
    min_r, grid = O.scatter_min_range(iRow * 5 + iCol, r, np.ones((5, 3), np.float32), 25, (5, 5, 3))
    assert min_r.tolist() == [7.0, 7.0, 7.0, 9.0, 8.0]
    want = np.zeros((5, 5, 3), np.float32)
    want[1, 1], want[1, 3], want[4, 4] = 2.0, 1.0, 1.0
    assert np.array_equal(grid, want)
