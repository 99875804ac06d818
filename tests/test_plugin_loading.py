"""The reference loads its renderer/network with imp.load_source(module_name, path) and then calls
.Renderer(network) / .Network() (make_renderer.py:4-8, make_network.py:4-8).  Check that our two
plugin files survive exactly that call (no package context), on the CPU."""
import os
import sys
import warnings

from conftest import REPO


def _load_source(name, path):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import imp                     # Python 3.10 (this image); the reference itself uses it
        return imp.load_source(name, path)


def test_loads_like_make_network_and_make_renderer(synthetic_sd):
    cwd = os.getcwd()
    os.chdir(REPO)                     # *_path is relative to the CWD (config.py:172-174)
    try:
        net_mod = _load_source("nerf_replication_amd.network", "nerf_replication_amd/network.py")
        ren_mod = _load_source("nerf_replication_amd.volume_renderer", "nerf_replication_amd/volume_renderer.py")
    finally:
        os.chdir(cwd)
    net = net_mod.Network()
    net.load_state_dict(synthetic_sd, strict=True)
    ren = ren_mod.Renderer(net)
    assert (ren.N_samples, ren.N_importance, ren.white_bkgd, ren.perturb) == (64, 128, True, False)
    assert net.chunk == 512 and net.N_samples == 64 and net.N_importance == 128
    assert hasattr(net, "embed_fn") and hasattr(net, "embeddirs_fn") and hasattr(net, "model_fine")
    assert len(list(net.named_parameters())) == 48


def test_overlay_shims_reexport_the_plugin_classes():
    sys.path.insert(0, os.path.join(REPO, "integration", "overlay"))
    try:
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
            del sys.modules[k]
        import importlib.util
        for rel, names in (("src/models/nerf/renderer/volume_renderer.py", ["Renderer"]),
                           ("src/models/nerf/network.py", ["Network", "NeRF"])):
            spec = importlib.util.spec_from_file_location("overlay_" + names[0],
                                                          os.path.join(REPO, "integration", "overlay", rel))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            for n in names:
                assert hasattr(mod, n)
    finally:
        sys.path.pop(0)
