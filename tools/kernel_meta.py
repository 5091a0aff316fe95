"""Read the code-object metadata of librodent_hip.so's gfx950 kernels: registers, spills, scratch, LDS and the byte offsets
the device compiler assigned to the kernel arguments.  Used by tests/test_abi_and_oracle.py (kernarg layout cross-check
against rr_kernarg_layout) and for the DESIGN.md register / scratch table.

usage: python tools/kernel_meta.py [path/to/librodent_hip.so]      (prints one line per kernel)
"""
import os
import shutil
import subprocess
import sys
import tempfile

import yaml

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(so_path):
    tmp = tempfile.mkdtemp(prefix="rr_meta_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(so_path, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=tmp)
        cos = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not cos:
            raise RuntimeError("no gfx950 code object in " + so_path)
        out = []
        for co in cos:
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co)], check=True,
                                   capture_output=True, text=True).stdout
            i = notes.index("amdhsa.kernels")
            y = notes[notes.rindex("---", 0, i):]
            y = y[:y.index("\n...")] if "\n..." in y else y
            for k in yaml.safe_load(y)["amdhsa.kernels"]:
                out.append(dict(name=k[".name"], vgpr=k[".vgpr_count"], agpr=k.get(".agpr_count", 0), sgpr=k[".sgpr_count"],
                                vgpr_spill=k.get(".vgpr_spill_count", 0), sgpr_spill=k.get(".sgpr_spill_count", 0),
                                scratch=k[".private_segment_fixed_size"], lds=k[".group_segment_fixed_size"],
                                kernarg_size=k[".kernarg_segment_size"],
                                args=[(a[".offset"], a[".size"], a.get(".value_kind")) for a in k[".args"]]))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "brax-rodent-run_amd", "csrc", "librodent_hip.so")
    for k in kernels(so):
        print(f"{k['name'][:70]:70s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} spill v{k['vgpr_spill']:3d} s{k['sgpr_spill']:3d} "
              f"scratch {k['scratch']:4d} B/lane  kernarg {k['kernarg_size']}  explicit args {[a[:2] for a in k['args'] if a[2] == 'by_value']}")
