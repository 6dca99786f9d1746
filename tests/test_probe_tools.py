"""The diagnostic generators under scripts/ must keep applying to the product's sources (they patch named lines of
csrc/kernels_graph.hip and assert that every anchor occurs exactly once): CPU only, nothing is compiled."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_k1_knockout_variants_still_apply(tmp_path):
    spec = importlib.util.spec_from_file_location("k1_variants", os.path.join(ROOT, "scripts", "probe", "k1_ab", "make_variants.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert {"zero", "nomfma", "halfmfma", "noepi", "noload", "sameload", "load32", "ldsload", "touch"} <= set(mod.VARIANTS)
    for name, make in mod.VARIANTS.items():
        text = make()
        assert text != mod.SRC and "tim_graph_mfma3_kernel" in text, name
        if name not in ("touch", "seq3", "seq4"):  # every variant with wrong bits forces its outputs to zero and drops its flags
            assert "ownw = 0;" in text and "unsigned int vf = 0;" in text, name
