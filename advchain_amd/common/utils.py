"""Model-state context managers used by the solver (reference: advchain/common/utils.py:114-173) and
the chain-sampling helper of the README usage (utils.py:180-212)."""
import contextlib
import random

import numpy as np
import torch

from .layers import Fixable2DDropout, Fixable3DDropout


def _toggle_fixable_dropout(model):
    for _, module in model.named_modules():
        if isinstance(module, (Fixable2DDropout, Fixable3DDropout)):
            module.lazy_load = not module.lazy_load


@contextlib.contextmanager
def _disable_tracking_bn_stats(model):
    """BatchNorm running statistics are not updated inside the block; Fixable*Dropout.lazy_load is
    flipped on entry and on exit (utils.py:114-147)."""
    saved = {}
    for name, module in model.named_modules():
        if isinstance(module, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            saved[name] = module.track_running_stats
            module.track_running_stats = False
    _toggle_fixable_dropout(model)
    yield
    for name, module in model.named_modules():
        if name in saved:
            module.track_running_stats = saved[name]
    _toggle_fixable_dropout(model)


@contextlib.contextmanager
def _fix_dropout(model):
    """Flip Fixable*Dropout.lazy_load inside the block (utils.py:149-173)."""
    _toggle_fixable_dropout(model)
    yield
    _toggle_fixable_dropout(model)


def set_grad(module, requires_grad=False):
    for p in module.parameters():
        p.requires_grad = requires_grad


def _shuffle_with_constant(x, r):
    """``random.shuffle(x, lambda: r)`` of Python <= 3.10 (the 2-argument form the reference calls, removed in 3.11):
    Fisher-Yates from the end with every draw equal to r, in place."""
    for i in reversed(range(1, len(x))):
        j = int(r * (i + 1))
        x[i], x[j] = x[j], x[i]


def random_chain(alist, max_length=None, size_list=None):
    """Random sub-chain in random order (utils.py:180-212), draw for draw like the reference: the length from
    ``np.random.randint``, ONE ``random.random()`` that drives the shuffle of ``alist`` (and of ``size_list`` with the
    same permutation), both shuffled IN PLACE as the reference does.  The reference raises NameError for one-element
    lists (undefined ``args``, utils.py:194); here that case returns the element."""
    length = len(alist)
    assert length >= 1, "input list must contains at least one element"
    max_length = length if max_length is None else min(max_length, length)
    if length == 1:
        return [alist[0]] if size_list is None else ([alist[0]], [size_list[0]])
    sub_len = np.random.randint(low=1, high=max_length + 1)
    r = random.random()
    _shuffle_with_constant(alist, r)
    if size_list is not None and len(size_list) >= 0:
        _shuffle_with_constant(size_list, r)
        return alist[:sub_len], size_list[:sub_len]
    return alist[:sub_len]


# ---------------------------------------------------------------------------------------------
# Data loading of the README / notebook usage (utils.py:13-96).  The reference reads volumes through SimpleITK; here a
# small reader for the NRRD files its example data ships in (attached data, raw or gzip encoding) keeps the package
# free of that dependency.  Array axes follow sitk.GetArrayFromImage: the FASTEST file axis is the LAST array axis
# (file sizes "228 271 10" -> array (10, 271, 228) = [D, H, W]).
# ---------------------------------------------------------------------------------------------
_NRRD_TYPES = {
    "signed char": "i1", "int8": "i1", "int8_t": "i1", "uchar": "u1", "unsigned char": "u1", "uint8": "u1", "uint8_t": "u1",
    "short": "i2", "short int": "i2", "signed short": "i2", "signed short int": "i2", "int16": "i2", "int16_t": "i2",
    "ushort": "u2", "unsigned short": "u2", "unsigned short int": "u2", "uint16": "u2", "uint16_t": "u2",
    "int": "i4", "signed int": "i4", "int32": "i4", "int32_t": "i4", "uint": "u4", "unsigned int": "u4", "uint32": "u4",
    "uint32_t": "u4", "longlong": "i8", "long long": "i8", "long long int": "i8", "signed long long": "i8", "int64": "i8",
    "int64_t": "i8", "ulonglong": "u8", "unsigned long long": "u8", "uint64": "u8", "uint64_t": "u8",
    "float": "f4", "double": "f8",
}


def read_nrrd(path):
    """Returns (array, header dict) of an NRRD file with attached data ('raw' or 'gzip'/'gz' encoding)."""
    import gzip
    with open(path, "rb") as f:
        blob = f.read()
    if not blob.startswith(b"NRRD"):
        raise ValueError("%s is not an NRRD file" % path)
    end = blob.find(b"\n\n")
    sep = 2
    crlf = blob.find(b"\r\n\r\n")
    if crlf >= 0 and (end < 0 or crlf < end):
        end, sep = crlf, 4
    if end < 0:
        raise ValueError("%s: NRRD header is not terminated by an empty line" % path)
    header = {}
    for line in blob[:end].decode("ascii", "replace").splitlines()[1:]:
        if not line or line.startswith("#"):
            continue
        key, colon, val = line.partition(":")
        if colon:
            header[key.strip().lower()] = val.lstrip("=").strip()
    if "data file" in header or "datafile" in header:
        raise NotImplementedError("%s: detached NRRD data files are not supported" % path)
    kind = header["type"].lower()
    if kind not in _NRRD_TYPES:
        raise NotImplementedError("%s: NRRD type '%s'" % (path, header["type"]))
    dtype = np.dtype(_NRRD_TYPES[kind])
    if dtype.itemsize > 1:
        dtype = dtype.newbyteorder("<" if header.get("endian", "little").lower() == "little" else ">")
    sizes = [int(s) for s in header["sizes"].split()]
    assert len(sizes) == int(header.get("dimension", len(sizes))), "NRRD 'sizes' does not match 'dimension'"
    payload = blob[end + sep:]
    encoding = header.get("encoding", "raw").lower()
    if encoding in ("gzip", "gz"):
        payload = gzip.decompress(payload)
    elif encoding != "raw":
        raise NotImplementedError("%s: NRRD encoding '%s'" % (path, header["encoding"]))
    count = int(np.prod(sizes))
    if len(payload) < count * dtype.itemsize:
        raise ValueError("%s: NRRD payload is shorter than its header says" % path)
    arr = np.frombuffer(payload, dtype=dtype, count=count).reshape(sizes[::-1])
    return arr.astype(dtype.newbyteorder("=")), header


def check_dir(dir_path, create=False):
    """1 if the directory exists, else -1 (created first when `create`)  (utils.py:13-26)."""
    import os
    if os.path.exists(dir_path):
        return 1
    if create:
        os.makedirs(dir_path)
    return -1


def load_image_label(image_path, label_path=None, slice_id=0, crop_size=(192, 192)):
    """Image (and optional label) volume from disk, one slice (slice_id >= 0) or the whole stack (slice_id < 0),
    centre-cropped to `crop_size` in-plane, image min-max rescaled to [0, 1]  (utils.py:29-80; .nrrd only here)."""
    from pathlib import Path
    assert Path(image_path).suffix == ".nrrd", "only .nrrd volumes can be read without SimpleITK"

    def cut(vol):
        if slice_id >= 0:
            vol = vol[slice_id]
        h_ind = 0 if slice_id >= 0 else 1
        h_diff = (vol.shape[h_ind] - crop_size[0]) // 2
        w_diff = (vol.shape[h_ind + 1] - crop_size[1]) // 2
        return vol, vol[..., h_diff:crop_size[0] + h_diff, w_diff:crop_size[1] + w_diff]

    image, cropped_image = cut(read_nrrd(image_path)[0])
    cropped_image = (cropped_image - cropped_image.min()) / (cropped_image.max() - cropped_image.min() + 1e-10)
    if label_path is None:
        return cropped_image
    label, cropped_label = cut(read_nrrd(label_path)[0])
    assert image.shape == label.shape, "The sizes of the input image and label do not match, image:{}label:{}".format(
        str(image.shape), str(label.shape))
    return cropped_image, cropped_label


def rescale_intensity(data, new_min=0, new_max=1, eps=1e-20):
    """Per (sample, channel) min-max rescaling of an N x C x H x W batch  (utils.py:82-95)."""
    bs, c, h, w = data.size(0), data.size(1), data.size(2), data.size(3)
    flat = data.reshape(bs * c, -1)
    old_max = torch.max(flat, dim=1, keepdim=True).values
    old_min = torch.min(flat, dim=1, keepdim=True).values
    return ((flat - old_min) / (old_max - old_min + eps) * (new_max - new_min) + new_min).view(bs, c, h, w)
