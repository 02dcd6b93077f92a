import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # tests that hand torch device tensors to the C ABI: torch's HIP runtime must come up before the library's first use of the
    # GPU in this process (afterwards torch reports "No HIP GPUs are available")
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def fixture_tiles(f):
    """Tile layout of a golden fixture: (1, 1) for the untiled runs, (columns, rows) for the uniformly spaced ones (rd_t*), explicit CTU
    sizes ([widths], [heights]) for the runs with TileUniformSpacing 0 (rd_n*)."""
    if "tile_col_sizes" in f.files:
        return [int(v) for v in f["tile_col_sizes"]], [int(v) for v in f["tile_row_sizes"]]
    return tuple(int(v) for v in f["tiles"]) if "tiles" in f.files else (1, 1)


def fixture_lf_offsets(f):
    """(LoopFilterBetaOffset_div2, LoopFilterTcOffset_div2) of a golden fixture (rd_o*: reference runs with the keys on the command line)."""
    return tuple(int(v) for v in f["lf_offsets"]) if "lf_offsets" in f.files else (0, 0)


def fixture_wavefront(f):
    """WaveFrontSynchro of a golden fixture (rd_w*: reference runs with --WaveFrontSynchro=1)."""
    return bool(int(f["wavefront"])) if "wavefront" in f.files else False


def fixture_lf(f):
    """LFCrossTileBoundaryFlag of a golden fixture (rd_l*: 0)."""
    return bool(int(f["lf_across_tiles"])) if "lf_across_tiles" in f.files else True


@pytest.fixture(scope="session")
def oracle_built():
    """Compile the plain-C oracle (test infrastructure) once per session."""
    import __graft_entry__ as g
    g.build_oracle()
    return True
