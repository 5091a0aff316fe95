"""Compiled model blobs shipped with the package.

The reference's MJCF files live only in /root/reference (absent on the GPU box), so the models are
compiled once where the XML is available (`build_assets`, called by __graft_entry__.build()) and the
resulting numeric tables (assets/*.rrm) are what the package loads at run time.
"""
from __future__ import annotations

import os

from . import mjcf

ASSET_DIR = os.environ.get("RR_ASSETS") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
MODELS = ("rodent_optimized", "rodent_new", "rodent_pair", "rodent_0", "rodent_cpu")
# rodent_cpu.xml (BASELINE config 1, CPU plumbing): ~4.3 k self-collision pairs of primitives that are not implemented are
# dropped at compile time (contacts off); its blob carries no kernel tables (hip_supported = 0)
_COMPILE_KW = {"rodent_cpu": dict(contacts="supported_only")}


def asset_path(name: str) -> str:
    return os.path.join(ASSET_DIR, f"{name}.rrm")


def _compiler_mtime() -> float:
    from . import ktables, levelsched
    return max(os.path.getmtime(mjcf.__file__), os.path.getmtime(ktables.__file__), os.path.getmtime(levelsched.__file__))


def build_assets(xml_dir: str, names=MODELS, force: bool = False):
    os.makedirs(ASSET_DIR, exist_ok=True)
    built = []
    for n in names:
        src, dst = os.path.join(xml_dir, f"{n}.xml"), asset_path(n)
        if not os.path.exists(src):
            continue
        if force or not os.path.exists(dst) or os.path.getmtime(dst) < _compiler_mtime():
            mjcf.save_blob(mjcf.compile_mjcf(src, **_COMPILE_KW.get(n, {})), dst)
            built.append(dst)
    return built


def resolve_model(xml_path: str) -> str:
    """Path of the compiled blob for `xml_path`: an .rrm file is used as is, an .xml is compiled
    next to the assets when it exists, otherwise the shipped blob of the same stem is used."""
    if xml_path.endswith(".rrm"):
        return xml_path
    stem = os.path.splitext(os.path.basename(xml_path))[0]
    if os.path.exists(xml_path):
        os.makedirs(ASSET_DIR, exist_ok=True)
        dst = asset_path(stem)
        if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(xml_path):
            mjcf.save_blob(mjcf.compile_mjcf(xml_path, **_COMPILE_KW.get(stem, {})), dst)
        return dst
    dst = asset_path(stem)
    if not os.path.exists(dst):
        raise FileNotFoundError(f"neither {xml_path} nor the compiled model {dst} exists")
    return dst
