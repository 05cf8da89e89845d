"""Short runs of the fuzzers of tools/ as part of the GPU suite (the long campaigns are run by hand):
  * tools/subset_fuzz.py -- random FAMILY SUBSETS x settings x batches mixing LDS-sized, wide and global-workspace ROIs, vs the oracle;
  * tools/tile_fuzz.py   -- the tile path (random label images, arbitrary label values, element types, chunk budgets) vs the batch
                            path on the same ROIs, bit for bit;
  * tools/ltex_fuzz.py   -- the several-workgroups-per-ROI texture path (boxes beyond LDS: thin, tall, wide, holes, flat patches, every
                            binning mode) vs the oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [1, 2])
def test_family_subset_fuzz(hip_ctx, seed):
    assert _tool("subset_fuzz").run(hip_ctx, seed=seed, rounds=8, seconds=60, verbose=False) == 0


@pytest.mark.parametrize("seed", [3, 4])
def test_tile_path_equals_batch_path_fuzz(hip_ctx, seed):
    assert _tool("tile_fuzz").run(hip_ctx, seed=seed, rounds=10, verbose=False) == 0


@pytest.mark.parametrize("seed", [5, 6])
def test_large_texture_path_fuzz(hip_ctx, seed):
    assert _tool("ltex_fuzz").run(hip_ctx, seed=seed, rounds=10, seconds=90, verbose=False) == 0
