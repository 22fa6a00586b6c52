"""CPU: the convolution kernel's dispatch-id -> logical-tile map (csrc/conv1d.hip: tile_of_workgroup, evaluated on
the host through pwg_debug_conv_tile_of_workgroup) and the planner's choice between its two orders (pwg_conv1d_plan).

The map decides only WHICH workgroup computes which tile, so any bijection gives identical results; what the two
orders change is which tiles share an XCD's L2.  The dispatcher deals consecutive workgroup ids round-robin to the
8 XCDs (MI355X_MICROARCH.md, workgroup dispatch), so XCD x runs the ids congruent to x modulo 8.
"""
import itertools
import os
import subprocess
import sys

import pytest

from parallelwavegan_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GRIDS = [  # (column tiles, row blocks, groups, items, slices)
    (1, 8, 1, 16, 4),   # 1024 -> 1024 k = 5 at T = 32 (scale discriminator tail), 128-row tiles, 4 slices
    (2, 12, 1, 16, 2),  # data gradient of a 1024 -> 512 stride-3 layer in polyphase form
    (3, 1, 16, 16, 1),  # grouped k = 41 layer, 16 groups
    (7, 3, 2, 5, 3),    # nothing divides anything
    (1, 1, 1, 1, 1),
    (5, 2, 1, 3, 1),    # fewer than 8 workgroups per XCD run boundary cases
    (25, 2, 1, 16, 1),  # generator-sized: many column tiles
]


def _walk(gx, mtiles, groups, items, ksplit, item_major):
    grid = (gx, mtiles * groups, items * ksplit)
    total = grid[0] * grid[1] * grid[2]
    return grid, [ops.conv_tile_of_workgroup(grid, mtiles, ksplit, item_major, wg) for wg in range(total)]


@pytest.mark.parametrize("shape", GRIDS)
@pytest.mark.parametrize("item_major", [False, True])
def test_tile_map_is_a_bijection(shape, item_major):
    gx, mtiles, groups, items, ksplit = shape
    grid, tiles = _walk(gx, mtiles, groups, items, ksplit, item_major)
    assert len(set(tiles)) == len(tiles) == grid[0] * grid[1] * grid[2]
    for bx, by, bz in tiles:
        assert 0 <= bx < grid[0] and 0 <= by < grid[1] and 0 <= bz < grid[2]


def _per_xcd(tiles, mtiles, ksplit):
    """Per XCD: the distinct weight tiles (row block, group, slice) and x windows (item, column tile, group, slice)."""
    weights, windows = [set() for _ in range(8)], [set() for _ in range(8)]
    for wg, (bx, by, bz) in enumerate(tiles):
        mi, grp, item, ks = by % mtiles, by // mtiles, bz // ksplit, bz % ksplit
        weights[wg % 8].add((mi, grp, ks))
        windows[wg % 8].add((item, bx, grp, ks))
    return weights, windows


def test_item_major_order_keeps_a_weight_tile_on_one_xcd():
    """1024 -> 1024, k = 5, 32 columns per item, B = 16, four reduction slices (512 workgroups): in the x-window-major
    order every XCD walks ALL 32 weight tiles of the layer, in the item-major order 4 of them -- each weight tile is
    streamed from HBM by exactly one XCD and shared there by the 16 items."""
    gx, mtiles, groups, items, ksplit = GRIDS[0]
    _, t0 = _walk(gx, mtiles, groups, items, ksplit, False)
    _, t1 = _walk(gx, mtiles, groups, items, ksplit, True)
    w0, x0 = _per_xcd(t0, mtiles, ksplit)
    w1, x1 = _per_xcd(t1, mtiles, ksplit)
    assert [len(s) for s in w0] == [32] * 8
    assert [len(s) for s in w1] == [4] * 8
    owners = {}
    for xcd, s in enumerate(w1):
        for t in s:
            assert owners.setdefault(t, xcd) == xcd
    assert len(owners) == mtiles * groups * ksplit
    # the price: an XCD now touches one slice of every item's x window (16 small windows instead of 8)
    assert [len(s) for s in x0] == [8] * 8 and [len(s) for s in x1] == [16] * 8


def test_x_window_major_order_keeps_a_column_tile_on_one_xcd():
    """Generator-sized launch (25 column tiles x 2 row blocks x 16 items): the row blocks of a column tile stay
    together, as they did before the second order existed."""
    gx, mtiles, groups, items, ksplit = GRIDS[6]
    _, tiles = _walk(gx, mtiles, groups, items, ksplit, False)
    by_xcd = [[] for _ in range(8)]
    for wg, t in enumerate(tiles):
        by_xcd[wg % 8].append(t)
    for run in by_xcd:
        for a, b in zip(run[0::2], run[1::2]):  # consecutive tiles of an XCD's run: the two row blocks of one window
            assert (a[0], a[2]) == (b[0], b[2]) and {a[1], b[1]} == {0, 1}


def _plan(batch, c_in, c_out, t_in, kernel, stride=1, dilation=1, groups=1, width=1, transposed=False, t_out=None):
    pad = (kernel - 1) // 2 * dilation
    if t_out is None:
        t_out = ops.conv_out_length(t_in, kernel, stride, dilation, pad, pad)
    d = ops.make_conv_desc(batch, c_in, c_out, t_in, t_out, kernel, stride, dilation, pad, groups,
                           transposed=transposed, width=width)
    return ops.conv1d_plan(d)


_PLANNER_CASES = """
import json, sys
from parallelwavegan_amd import ops
def plan(batch, c_in, c_out, t_in, kernel, stride=1, width=1, t_out=None):
    pad = (kernel - 1) // 2
    if t_out is None:
        t_out = ops.conv_out_length(t_in, kernel, stride, 1, pad, pad)
    return ops.conv1d_plan(ops.make_conv_desc(batch, c_in, c_out, t_in, t_out, kernel, stride, 1, pad, 1, width=width))
out = dict(tail=[plan(16, 1024, 1024, t, 5) for t in (9, 17, 32)],
           mpd=[plan(16, 1024, 1024, 21, 5, width=5), plan(16, 512, 1024, 61, 5, stride=3, width=5, t_out=21)],
           gen=[plan(16, c, c, t, k) for c, t, k in ((128, 51200, 11), (256, 6400, 7), (64, 102400, 3), (512, 800, 7))],
           one=[plan(1, 1024, 1024, 32, 5)])
print(json.dumps(out))
"""


def _planner_cases(order):
    import json

    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("PWG_TILE_ORDER", None)
    if order is not None:
        env["PWG_TILE_ORDER"] = str(order)
    out = subprocess.run([sys.executable, "-c", _PLANNER_CASES], env=env, capture_output=True, text=True, cwd=ROOT,
                         timeout=300)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().split("\n")[-1])


def test_traffic_model_picks_the_item_major_order_for_weight_heavy_layers_only():
    """PWG_TILE_ORDER=2 (the mode is read once per process: subprocesses): the bytes-per-XCD model takes the item-major
    order for the scale / period discriminators' 1024-channel layers at B = 16 (reference: models/hifigan.py:314-341,
    :529-601) and leaves the generator layers of the inference batch (x windows dominate) and single items alone."""
    r = _planner_cases(2)
    for p in r["tail"] + r["mpd"]:
        assert p["family"] == "mfma" and p["dma"] and p["item_major"], p
    for p in r["gen"] + r["one"]:
        assert p["family"] == "mfma" and not p["item_major"], p


def test_default_and_forced_tile_orders():
    """Default: the x-window-major order everywhere (the item-major one measured no faster, profiles/r05_tile_order_ab.txt);
    PWG_TILE_ORDER=1 forces item-major."""
    for order, want in ((None, False), (0, False), (1, True)):
        r = _planner_cases(order)
        for p in r["tail"] + r["mpd"] + r["gen"]:
            assert p["item_major"] == want, (order, p)


def test_plan_reports_the_other_kernel_families():
    assert _plan(16, 1, 128, 8192, 15)["family"] == "single_input_channel"
    assert _plan(16, 32, 1, 204800, 7)["family"] == "few_output_channels"
    p = _plan(16, 1024, 1024, 32, 5)
    assert p["grid"][0] * p["grid"][1] * p["grid"][2] >= 256 and p["ksplit"] >= 1


def test_plan_grid_matches_the_tile_map_domain():
    """Every planned launch's grid is a valid domain of the tile map (row blocks divide grid y, slices divide grid z)."""
    for b, c_in, c_out, t, k, s, g in itertools.product((1, 16), (64, 1024), (128, 1024), (9, 300), (3, 5), (1, 3), (1,)):
        d = ops.make_conv_desc(b, c_in, c_out, t, ops.conv_out_length(t, k, s, 1, k // 2, k // 2), k, s, 1, k // 2, g)
        p = ops.conv1d_plan(d)
        if p["family"] != "mfma":
            continue
        gx, gy, gz = p["grid"]
        assert gz == b * p["ksplit"] and gx >= 1 and gy >= 1
        last = gx * gy * gz - 1
        ops.conv_tile_of_workgroup(p["grid"], gy, p["ksplit"], p["item_major"], last)  # (groups = 1: row blocks = gy)
