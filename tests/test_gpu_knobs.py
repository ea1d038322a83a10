"""The diagnostic knobs (PXZ_* environment variables, read once per process: pxz_tables.cpp knobs()) select the older form
of a path -- the round-1 resample forms, the widened RGB copy, the generic kernel's staging of unaligned rows ... -- which
remains the fallback when the newer one does not apply.  Each is run once, in a child process of its own, over the parity
cases that reach it: a fallback that nothing exercises rots."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KNOBS = [
    ("PXZ_NO_NARROW", "test_every_filter_on_opaque_fast_path or test_rgb_frames_on_the_square_fast_paths"),
    # (the per-tile dot2 resamples of shrink16_kernel, which the block-diagonal matrix-core form of a whole group replaces by default)
    ("PXZ_NO_GROUP16", "test_block16_group_kernel_every_class or test_rgb_frames_on_the_square_fast_paths"),
    ("PXZ_NO_NATIVE_RGB", "test_rgb_frames_on_the_square_fast_paths or test_strided_batches_on_the_fast_paths"),
    ("PXZ_NO_REPITCH", "test_pitch_and_unaligned_rows or test_strided_batches_on_the_fast_paths"),
    # (the generic kernel on RGB: without the widening, large RGB tiles in shrink_by do not fit LDS -- documented -- so only the
    # cases that do)
    ("PXZ_NO_WIDEN", "test_process_matches_oracle or test_rgb_frames_on_the_square_fast_paths"),
    ("PXZ_OKLAB_V1", "test_shrink_1080p_rgba_32 or test_shrink_by_blocks_16_and_64"),
    # (shrink_by without the detector's copy of every tile into its slot: the shrink kernel reads and clones the tiles stored at full size itself)
    ("PXZ_NO_CLONE_AHEAD", "test_shrink_1080p_rgba_32 or test_shrink_by_blocks_16_and_64 or test_transparent_tiles_through_the_alpha_kernel"),
    # (the per-level grids and the rectangle lists are two implementations of the same recursion: where both apply, both run)
    ("PXZ_TREE_RECTS", "test_tree_process_matches_oracle or test_tree_process_edge_cases"),
    # (expand_kernel's general forms for 32x32 RGBA tiles, which the matrix-core / shift-indexed forms replace by default)
    ("PXZ_NO_EXPAND_FAST32", "test_expand_matches_oracle or test_expand_of_power_of_two_tiles or test_process_matches_oracle"),
    ("PXZ_NO_ALPHA_FIRST", "test_mostly_transparent_batches_go_to_the_four_plane_kernel_first or test_transparent_tiles_through_the_alpha_kernel"),
    ("PXZ_NO_ALPHA_KERNEL", "test_transparent_tiles_through_the_alpha_kernel or test_transparent_64x64_tiles_through_the_alpha_instance"),
]


@pytest.mark.parametrize("knob,cases", KNOBS)
def test_fallback_paths_behind_the_knobs(knob, cases):
    env = dict(os.environ)
    env[knob] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
                        "-k", cases, "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, f"{knob}=1:\n{tail}\n{r.stderr[-2000:]}"
    assert " passed" in tail and " failed" not in tail, tail
