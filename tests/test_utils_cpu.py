"""Model-side helpers of the reference's README usage (common/utils.py:114-212, common/layers.py:5-63): host logic only."""
import random

import pytest

import numpy as np
import torch

from advchain_amd.common.layers import Fixable2DDropout, Fixable3DDropout
from advchain_amd.common.utils import _disable_tracking_bn_stats, _fix_dropout, random_chain, set_grad


def test_fixable_dropout_replays_its_mask_only_when_asked():
    x = torch.ones(2, 8, 5, 5)
    drop = Fixable2DDropout(p=0.5, lazy_load=True)
    a = drop(x)          # first call: draws and stores a seed
    b = drop(x)          # lazy_load: same channels dropped again
    assert torch.equal(a, b)
    drop.lazy_load = False
    outs = [drop(x) for _ in range(6)]
    assert any(not torch.equal(a, o) for o in outs)       # fresh masks
    drop.eval()
    assert torch.equal(drop(x), x)
    assert Fixable3DDropout(p=0.3, lazy_load=True)(torch.ones(1, 4, 3, 3, 3)).shape == (1, 4, 3, 3, 3)


def test_bn_tracking_and_dropout_toggles_are_restored():
    model = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3, 1, 1), torch.nn.BatchNorm2d(4), Fixable2DDropout(0.2))
    bn, drop = model[1], model[2]
    assert bn.track_running_stats and not drop.lazy_load
    with _disable_tracking_bn_stats(model):
        assert not bn.track_running_stats and drop.lazy_load
        before = bn.running_mean.clone()
        model(torch.rand(2, 1, 6, 6))
        assert torch.equal(bn.running_mean, before)        # statistics frozen inside the block
    assert bn.track_running_stats and not drop.lazy_load
    with _fix_dropout(model):
        assert drop.lazy_load
    assert not drop.lazy_load
    set_grad(model, False)
    assert all(not p.requires_grad for p in model.parameters())


def test_random_chain_is_a_shuffled_prefix_with_matching_sizes():
    random.seed(0)
    np.random.seed(0)
    items, sizes = ["noise", "bias", "morph", "affine"], [1, 2, 3, 4]
    seen = set()
    for _ in range(50):
        chain, sz = random_chain(items, max_length=3, size_list=sizes)
        assert 1 <= len(chain) <= 3 and len(set(chain)) == len(chain)
        assert sz == [sizes[items.index(c)] for c in chain]
        seen.add(tuple(chain))
    assert len(seen) > 5
    assert random_chain(["only"]) == ["only"]
    assert random_chain(["only"], size_list=[7]) == (["only"], [7])


def _write_nrrd(path, arr, encoding="raw", type_name=None, endian="little"):
    import gzip
    names = {"float32": "float", "int16": "short", "uint8": "unsigned char", "float64": "double"}
    body = arr.astype(arr.dtype.newbyteorder("<" if endian == "little" else ">")).tobytes()
    if encoding == "gzip":
        body = gzip.compress(body)
    head = ("NRRD0004\n# comment line\ntype: %s\ndimension: %d\nspace: left-posterior-superior\nsizes: %s\n"
            "kinds: %s\nendian: %s\nencoding: %s\n\n") % (type_name or names[str(arr.dtype)], arr.ndim,
                                                          " ".join(str(s) for s in arr.shape[::-1]),
                                                          " ".join(["domain"] * arr.ndim), endian, encoding)
    with open(path, "wb") as f:
        f.write(head.encode("ascii") + body)


def test_nrrd_reader_and_image_loading(tmp_path):
    """Caller utilities of the README usage (utils.py:13-96) without SimpleITK: NRRD (raw / gzip, both byte orders),
    slice + centre crop + min-max rescale, per-sample intensity rescaling."""
    from advchain_amd.common.utils import check_dir, load_image_label, read_nrrd, rescale_intensity
    rng = np.random.default_rng(0)
    vol = (rng.random((10, 27, 22)) * 900 - 100).astype(np.float32)          # [D, H, W] like the example volume (x fastest)
    lab = (rng.random((10, 27, 22)) * 4).astype(np.int16)
    _write_nrrd(tmp_path / "img.nrrd", vol)
    _write_nrrd(tmp_path / "img_be.nrrd", vol, encoding="gzip", endian="big")
    _write_nrrd(tmp_path / "lab.nrrd", lab, encoding="gzip")
    a, hdr = read_nrrd(str(tmp_path / "img.nrrd"))
    assert a.shape == (10, 27, 22) and a.dtype == np.float32 and np.array_equal(a, vol)
    assert hdr["sizes"] == "22 27 10" and hdr["encoding"] == "raw"
    assert np.array_equal(read_nrrd(str(tmp_path / "img_be.nrrd"))[0], vol)
    assert np.array_equal(read_nrrd(str(tmp_path / "lab.nrrd"))[0], lab)
    img, seg = load_image_label(str(tmp_path / "img.nrrd"), str(tmp_path / "lab.nrrd"), slice_id=5, crop_size=(16, 12))
    ref = vol[5][5:21, 5:17]
    assert img.shape == (16, 12) and np.allclose(img, (ref - ref.min()) / (ref.max() - ref.min() + 1e-10))
    assert np.array_equal(seg, lab[5][5:21, 5:17])
    stack = load_image_label(str(tmp_path / "img.nrrd"), slice_id=-1, crop_size=(16, 12))
    assert stack.shape == (10, 16, 12) and float(stack.min()) == 0.0 and abs(float(stack.max()) - 1.0) < 1e-6
    x = torch.rand(3, 2, 5, 7) * 40 - 7
    y = rescale_intensity(x, 0, 1)
    assert y.shape == x.shape and torch.allclose(y.amin(dim=(2, 3)), torch.zeros(3, 2)) and torch.allclose(y.amax(dim=(2, 3)), torch.ones(3, 2))
    assert check_dir(str(tmp_path)) == 1 and check_dir(str(tmp_path / "new"), create=True) == -1 and check_dir(str(tmp_path / "new")) == 1
    with pytest.raises(ValueError):
        (tmp_path / "bad.nrrd").write_bytes(b"not a volume")
        read_nrrd(str(tmp_path / "bad.nrrd"))


def test_caller_utilities_match_the_reference_golden():
    """G9: random_chain draw for draw under fixed seeds (incl. the in-place shuffle of its
    arguments), rescale_intensity, and -- when the reference checkout is mounted -- the NRRD reader and load_image_label
    on the reference's own example volume against an independent decode of that file."""
    import os
    import random
    import hashlib
    import numpy as np
    import torch
    from tests.helpers import Fixture
    from advchain_amd.common import utils as U
    fx = Fixture("g9_utils")
    meta = fx.json()
    for c in meta["random_chain"]:
        np.random.seed(c["seed"])
        random.seed(1000 + c["seed"])
        a, s = list(c["names"]), list(c["sizes"])
        res = U.random_chain(a, max_length=c["max_length"], size_list=s if c["with_sizes"] else None)
        if c["with_sizes"]:
            assert list(res[0]) == c["result"] and list(res[1]) == c["result_sizes"], c["seed"]
            assert s == c["sizes_after"]
        else:
            assert list(res) == c["result"], c["seed"]
        assert a == c["alist_after"]
    out = U.rescale_intensity(fx.t("rescale_in").clone(), new_min=-1, new_max=2)
    assert float((out - fx.t("rescale_out")).abs().max()) < 1e-6
    path = os.path.join(os.environ.get("ADVCHAIN_REFERENCE_ROOT", "/root/reference"), meta["nrrd"]["relative_path"])
    if not os.path.exists(path):
        return                      # the reference's data does not travel to the GPU box
    vol, header = U.read_nrrd(path)
    m = meta["nrrd"]
    assert list(vol.shape) == m["shape"] and str(vol.dtype) == m["dtype"]
    assert hashlib.sha256(np.ascontiguousarray(vol).tobytes()).hexdigest() == m["sha256"]
    assert np.array_equal(vol[0, 40:56, 60:76], fx.arr("nrrd_slice0_patch"))
    img = U.load_image_label(path, slice_id=0, crop_size=(192, 192))
    assert img.shape == (192, 192) and abs(float(img.astype(np.float64).sum()) - m["loaded_sum"]) < 1e-6 * m["loaded_sum"]
    assert np.allclose(img[::16, ::16], fx.arr("nrrd_loaded_sample"), atol=1e-7)


def test_profile_tooling_recomputes_the_committed_numbers(tmp_path):
    """The evidence tooling on the committed round-4 profiles: tools/traffic_from_pmc.py rebuilds traffic.json (HBM bytes per
    launch from the PMC passes, duration and roofline fraction from the kernel trace of the same command) and
    tools/ns_pair_summary.py the north-star pair's fraction -- the numbers DESIGN.md quotes must come out of the CSVs."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles", "r04")
    out = str(tmp_path / "traffic.json")
    subprocess.run([sys.executable, os.path.join(root, "tools", "traffic_from_pmc.py"), prof, out], check=True, stdout=subprocess.DEVNULL)
    t = json.load(open(out))
    committed = json.load(open(os.path.join(prof, "traffic.json")))
    for wl in ("cfg2", "cfg3", "cfg5"):
        rec = t[wl]["advchain_compose_self_bwd"]
        assert rec == committed[wl]["advchain_compose_self_bwd"]
        assert 0.2 < rec["frac_from_rocprof"] < 0.45 and 1.0 < rec["traffic_over_algorithmic"] < 1.6
    assert abs(t["cfg2"]["advchain_compose_self_bwd"]["frac_from_rocprof"] - 0.292) < 0.002
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ns_pair_summary.py"), prof], check=True, capture_output=True, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("level")]
    assert len(lines) == 2 and "0.586 of" in lines[0] and "0.314 of" in lines[1], r.stdout


def test_committed_rocprof_summaries_are_consistent_with_their_own_runs():
    """profiles/rNN/per_call_summary.txt: the per-call GPU-busy time of a workload cannot exceed the ms_per_step the SAME
    profiled command printed (round 4 divided cfg-5's kernel time by one call too few)."""
    import glob
    import json
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    checked = 0
    for summary in sorted(glob.glob(os.path.join(root, "r*", "per_call_summary.txt"))):
        rnd = os.path.dirname(summary)
        if int(os.path.basename(rnd)[1:]) < 4:
            continue                # (rounds 1-3 profiled cold processes: MIOpen's find kernels sit inside their totals)
        for m in re.finditer(r"(\w+)_kernel_stats\.csv: GPU busy ([0-9.]+) ms/call", open(summary).read()):
            log = os.path.join(rnd, m.group(1) + "_under_rocprof.log")
            if not os.path.exists(log):
                continue
            rec = None
            for line in open(log, errors="replace"):
                if line.startswith("{"):
                    rec = json.loads(line)
                    break
            if rec is None or "ms_per_step" not in rec:
                continue            # (kernel_bench runs print no bench line)
            assert float(m.group(2)) <= 1.02 * rec["ms_per_step"], (summary, m.group(1), m.group(2), rec["ms_per_step"])
            checked += 1
    assert checked >= 3
