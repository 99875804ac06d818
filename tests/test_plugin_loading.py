"""The reference loads its renderer/network with imp.load_source(module_name, path) and then calls
.Renderer(network) / .Network() (make_renderer.py:4-8, make_network.py:4-8).  These tests run that call in a
NEW interpreter (nothing of the build pre-imported, so import order inside the test session cannot mask a
circular import -- round-1 VERDICT "Weak 1"), from a directory that holds nothing but a symlink to the package,
i.e. the layout INTEGRATION.md section 1 documents.  CPU only: construction and state_dict handling are host logic."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import GOLDEN, REPO

_LOADER = textwrap.dedent("""
    import os, sys, warnings
    warnings.simplefilter("ignore")
    import imp                                             # Python 3.10 (this image); the reference itself uses it
    assert not any(m.startswith("nerf_replication_amd") for m in sys.modules)
    {path_setup}
    net_mod = imp.load_source("nerf_replication_amd.network", "nerf_replication_amd/network.py")
    net = net_mod.Network()
    ren_mod = imp.load_source("nerf_replication_amd.volume_renderer", "nerf_replication_amd/volume_renderer.py")
    ren = ren_mod.Renderer(net)
    import torch
    ck = torch.load({ckpt!r}, weights_only=True)
    net.load_state_dict(ck["net"], strict=True)
    assert (ren.N_samples, ren.N_importance, ren.white_bkgd, ren.perturb) == (64, 128, True, False)
    assert net.chunk == 512 and net.N_samples == 64 and net.N_importance == 128
    assert hasattr(net, "embed_fn") and hasattr(net, "embeddirs_fn") and hasattr(net, "model_fine")
    assert len(list(net.named_parameters())) == 48
    import nerf_replication_amd as pkg                     # the lazy package attributes resolve to the SAME classes
    assert pkg.Network is net_mod.Network and pkg.Renderer is ren_mod.Renderer
    print("PLUGIN-OK")
""")


def _run(code, cwd):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    return subprocess.run([sys.executable, "-c", code], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("cwd_on_path", [True, False], ids=["cwd-on-sys.path", "cwd-not-on-sys.path"])
def test_loads_like_make_network_and_make_renderer_in_a_fresh_interpreter(tmp_path, cwd_on_path):
    os.symlink(os.path.join(REPO, "nerf_replication_amd"), tmp_path / "nerf_replication_amd")
    # `python run.py` puts the script's directory (= the reference root = the CWD) first on sys.path; a harness that
    # imports run_evaluate from elsewhere does not -- the plugin files then add the package's parent themselves
    setup = "sys.path.insert(0, os.getcwd())" if cwd_on_path else \
        "sys.path[:] = [p for p in sys.path if p not in ('', os.getcwd())]"
    res = _run(_LOADER.format(path_setup=setup, ckpt=os.path.join(GOLDEN, "synthetic_ckpt.pth")), str(tmp_path))
    assert res.returncode == 0 and "PLUGIN-OK" in res.stdout, res.stdout + res.stderr


def test_package_import_is_lazy():
    """`import nerf_replication_amd` must not import .network / .volume_renderer (that eager import was the cycle)."""
    code = ("import sys; sys.path.insert(0, %r); import nerf_replication_amd as p; "
            "assert 'nerf_replication_amd.network' not in sys.modules; "
            "assert 'nerf_replication_amd.volume_renderer' not in sys.modules; "
            "p.Network, p.Renderer, p.NeRF, p.Evaluator, p.generate_rays, p.load_network, p._lib; print('LAZY-OK')" % REPO)
    res = _run(code, "/tmp")
    assert res.returncode == 0 and "LAZY-OK" in res.stdout, res.stdout + res.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="build container only: needs the reference checkout")
def test_reference_make_network_and_make_renderer_load_the_build():
    """oracle/check_dropin.py: the reference's OWN src.config + make_network(cfg) + make_renderer(cfg, net) on the
    YAML of INTEGRATION.md section 1, 48 keys strict-loaded from the reference's Network."""
    res = subprocess.run([sys.executable, os.path.join(REPO, "oracle", "check_dropin.py")], cwd="/tmp",
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "make_network/make_renderer loaded the build" in res.stdout, res.stdout + res.stderr


def test_overlay_shims_reexport_the_plugin_classes(tmp_path):
    """integration/overlay/src/...: train.py's NetworkWrapper imports src.models.nerf.renderer.volume_renderer.Renderer
    by hard path (trainers/nerf.py:3); the overlay files re-export our classes under those paths."""
    code = textwrap.dedent("""
        import importlib.util, os, sys
        sys.path.insert(0, {repo!r})
        for rel, names in (("src/models/nerf/renderer/volume_renderer.py", ["Renderer"]),
                           ("src/models/nerf/network.py", ["Network", "NeRF"])):
            spec = importlib.util.spec_from_file_location("overlay_" + names[0], os.path.join({repo!r}, "integration", "overlay", rel))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            for n in names:
                assert hasattr(mod, n)
        print("OVERLAY-OK")
    """).format(repo=REPO)
    res = _run(code, str(tmp_path))
    assert res.returncode == 0 and "OVERLAY-OK" in res.stdout, res.stdout + res.stderr
