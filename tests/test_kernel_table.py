"""CPU: DESIGN.md section 5's register table against the code objects the Makefile built (tools/kernel_resources.py --built: the
.hip_fatbin sections of csrc/build/*.hip.o, no compile).  A kernel of the table that drifts by more than 8 VGPRs, changes its waves per
SIMD or disappears fails the test: regenerate the table with `python tools/kernel_resources.py --update-design` and re-read what
DESIGN.md says about that kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_design_register_table_matches_the_built_kernels(built):
    import kernel_resources as K
    with open(os.path.join(ROOT, "DESIGN.md")) as f:
        doc = K.parse_design_table(f.read())
    tab = K.built_table()
    # the hot-path kernels of the single C2 graph, the large batch and C3 / C5 must be listed
    for must in ("k_linearize_lanes", "k_hblocks2", "k_band_factor_pre", "k_band_root<true, true>", "k_band_solve_flow", "k_band_factor_r5",
                 "kb_level_factor3", "kb_level_solve", "kb_hblocks_tc", "k_trial_dual"):
        assert must in doc, must
    for name, (vgpr, agpr, waves, sspill, vspill, code) in doc.items():
        assert name in tab, f"DESIGN.md lists {name}, the build has no such kernel"
        r = tab[name]
        assert abs(r["vgpr"] - vgpr) <= 8, (name, r["vgpr"], vgpr)
        assert r["waves"] == waves, (name, r["waves"], waves)
        assert (r["vspill"] > 0) == (vspill > 0), (name, r["vspill"], vspill)
