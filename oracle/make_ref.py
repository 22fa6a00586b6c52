"""Stage the UNMODIFIED reference package as ``oracle/_ref/`` (TEST INFRASTRUCTURE).

    python oracle/make_ref.py          # in the build container, where /root/reference exists

``oracle/_ref/`` is listed in .gitignore (the reference's sources never enter the history) but NOT in
.gpurunignore, so -- like the built ``.so`` files -- it travels to the GPU box with the snapshot.  There
``oracle.ref_shim`` resolves ``import parallel_wavegan`` to it, and ``bench.py``'s ``cpu_baseline`` leg times the
reference's OWN modules (``kind: "reference"``) instead of the restatement (``kind: "port"``).  Only what the hot
path imports is staged: the ``parallel_wavegan`` Python package (models / layers / losses / optimizers / utils /
bin / datasets / distributed), byte for byte, plus a MANIFEST with the sha256 of every staged file so a test can
prove that nothing was edited on the way.  Nothing under ``parallelwavegan_amd/`` may import it.

Round 5: the reference's own unit tests, ``test/*.py``, are staged next to it as
``oracle/_ref/test/`` (same manifest) -- ``tests/test_reference_unit_tests_gpu.py`` runs those files, unedited,
against the DROP-IN package (``parallelwavegan_amd.compat.install()``) on the GPU box.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_ROOT = os.environ.get("PWG_REFERENCE_ROOT", "/root/reference")
DST_ROOT = os.path.join(HERE, "_ref")


def _sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def stage(force=False):
    """Copy <reference>/parallel_wavegan/**/*.py -> oracle/_ref/parallel_wavegan/.  Returns the manifest dict, or
    None when the reference tree is absent (the GPU box: the staged copy from the snapshot is used as it is)."""
    src = os.path.join(SRC_ROOT, "parallel_wavegan")
    if not os.path.isdir(src):
        return None
    dst = os.path.join(DST_ROOT, "parallel_wavegan")
    manifest_path = os.path.join(DST_ROOT, "MANIFEST.json")
    files = {}
    for top in (src, os.path.join(SRC_ROOT, "test")):
        for base, _, names in os.walk(top):
            for n in sorted(names):
                if n.endswith(".py"):
                    p = os.path.join(base, n)
                    files[os.path.relpath(p, SRC_ROOT)] = _sha(p)
    # the recipes' training configurations (egs/*/*/conf/*.yaml): what a user of the reference would hand to
    # build_from_config; tests/test_reference_recipes_gpu.py runs one small training step of every one of them
    egs = os.path.join(SRC_ROOT, "egs")
    for base, _, names in os.walk(egs):
        if os.path.basename(base) == "conf":
            for n in sorted(names):
                if n.endswith(".yaml"):
                    p = os.path.join(base, n)
                    files[os.path.relpath(p, SRC_ROOT)] = _sha(p)
    if not force and os.path.exists(manifest_path):
        with open(manifest_path) as f:
            old = json.load(f)
        if old.get("files") == files and all(os.path.exists(os.path.join(DST_ROOT, r)) for r in files):
            return old
    shutil.rmtree(dst, ignore_errors=True)
    shutil.rmtree(os.path.join(DST_ROOT, "test"), ignore_errors=True)
    shutil.rmtree(os.path.join(DST_ROOT, "egs"), ignore_errors=True)
    for rel in files:
        out = os.path.join(DST_ROOT, rel)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        shutil.copyfile(os.path.join(SRC_ROOT, rel), out)
    manifest = {"source": SRC_ROOT, "files": files}
    with open(manifest_path, "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    return manifest


def verify():
    """True when every staged file still has the sha256 the manifest recorded at staging time."""
    manifest_path = os.path.join(DST_ROOT, "MANIFEST.json")
    if not os.path.exists(manifest_path):
        return False
    with open(manifest_path) as f:
        files = json.load(f)["files"]
    return all(os.path.exists(os.path.join(DST_ROOT, r)) and _sha(os.path.join(DST_ROOT, r)) == h for r, h in files.items())


if __name__ == "__main__":
    m = stage(force="--force" in sys.argv)
    print("oracle/_ref:", "reference tree absent, nothing staged" if m is None else f"{len(m['files'])} files staged, verify={verify()}")
