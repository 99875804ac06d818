"""Drop-in check through the REAL reference loader  --  test infrastructure, build container only.

Builds the layout INTEGRATION.md section 1 documents (the package symlinked at the reference root, the two
`*_module` keys of configs/nerf/lego.yaml pointed at it) in a scratch directory -- /root/reference itself is
read-only, so `src/` and `configs/` are symlinked next to the package -- and then, in THIS fresh interpreter
with nothing of the build imported beforehand, does what run.py:156/:164 do:

    from src.config import cfg                      # argparse-at-import, *_path derivation (config.py:172-174)
    network  = make_network(cfg)                    # src/models/make_network.py:4-8   (imp.load_source)
    renderer = make_renderer(cfg, network)          # src/models/nerf/renderer/make_renderer.py:4-8

and checks the result against the reference's own Network / Renderer built the same way from the untouched
lego.yaml keys: 48 state_dict keys, strict load in both directions, identical constructor-derived attributes.
No GPU and no compute call is needed (the product path has no CPU fallback; construction is host logic).

    python oracle/check_dropin.py          # exit 0 = the documented drop-in loads; skipped (exit 0) without /root/reference
"""
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    if not os.path.isdir(REF):
        print("check_dropin: /root/reference absent, skipped")
        return 0
    root = tempfile.mkdtemp(prefix="nerf_dropin_")
    for name in ("src", "configs"):
        os.symlink(os.path.join(REF, name), os.path.join(root, name))
    os.symlink(os.path.join(REPO, "nerf_replication_amd"), os.path.join(root, "nerf_replication_amd"))
    with open(os.path.join(REF, "configs/nerf/lego.yaml")) as f:
        yaml_text = f.read()
    swaps = (("network_module: src.models.nerf.network", "network_module: nerf_replication_amd.network"),
             ("renderer_module: src.models.nerf.renderer.volume_renderer",
              "renderer_module: nerf_replication_amd.volume_renderer"))
    for old, new in swaps:
        assert old in yaml_text, old
        yaml_text = yaml_text.replace(old, new)
    with open(os.path.join(root, "lego_mi355x.yaml"), "w") as f:
        f.write(yaml_text)

    # what `cd <reference root> && python run.py --type evaluate --cfg_file lego_mi355x.yaml` sets up
    os.chdir(root)
    sys.path[0:0] = [root]
    sys.argv = ["run.py", "--type", "evaluate", "--cfg_file", "lego_mi355x.yaml"]
    assert not any(m.startswith("nerf_replication_amd") for m in sys.modules), "build pre-imported: not a fresh process"

    from src.config import cfg
    from src.models import make_network
    from src.models.nerf.renderer import make_renderer
    assert cfg.network_path == "nerf_replication_amd/network.py", cfg.network_path
    assert cfg.renderer_path == "nerf_replication_amd/volume_renderer.py", cfg.renderer_path

    network = make_network(cfg)
    renderer = make_renderer(cfg, network)
    assert type(network).__module__ == "nerf_replication_amd.network", type(network).__module__
    assert type(renderer).__module__ == "nerf_replication_amd.volume_renderer", type(renderer).__module__
    assert renderer.net is network

    # the reference's own classes, through the same loader with the original module keys
    import imp
    ref_net = imp.load_source("src.models.nerf.network", "src/models/nerf/network.py").Network()
    ref_ren = imp.load_source("src.models.nerf.renderer.volume_renderer",
                              "src/models/nerf/renderer/volume_renderer.py").Renderer(ref_net)

    ref_sd = ref_net.state_dict()
    assert len(ref_sd) == 48 and list(network.state_dict().keys()) == list(ref_sd.keys())
    network.load_state_dict(ref_sd, strict=True)                 # net_utils.py:375 (reference checkpoint -> build)
    ref_net.load_state_dict(network.state_dict(), strict=True)   # and the other way round
    for k, v in ref_sd.items():
        assert network.state_dict()[k].shape == v.shape and (network.state_dict()[k] == v).all(), k
    assert [n for n, _ in network.named_parameters()] == [n for n, _ in ref_net.named_parameters()]

    for attr in ("N_samples", "N_importance", "white_bkgd", "perturb", "fast_sampling", "weights_threshold",
                 "rays_size", "sample_size", "chunk_size", "task"):
        assert getattr(renderer, attr) == getattr(ref_ren, attr), (attr, getattr(renderer, attr), getattr(ref_ren, attr))
    for attr in ("N_samples", "N_importance", "chunk", "batch_size", "white_bkgd", "use_viewdirs", "input_ch",
                 "input_ch_views"):
        assert getattr(network, attr) == getattr(ref_net, attr), (attr, getattr(network, attr), getattr(ref_net, attr))
    for attr in ("model", "model_fine", "embed_fn", "embeddirs_fn"):
        assert hasattr(network, attr), attr
    network.eval(); network.train(); network.to("cpu")
    print("check_dropin: make_network/make_renderer loaded the build; 48 keys strict-load both ways; "
          "constructor attributes equal the reference's")
    return 0


if __name__ == "__main__":
    sys.exit(main())
