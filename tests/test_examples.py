"""The programs under examples/ build (C, against the public header only) and run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_c(tmp_path):
    out = tmp_path / "resident_nmf"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "resident_nmf.c"), "-o", str(out),
                        "-L" + os.path.join(ROOT, "smallk_amd", "lib"), "-lsmallk_amd",
                        "-Wl,-rpath," + os.path.join(ROOT, "smallk_amd", "lib")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_c_example_compiles_as_c99(tmp_path):
    """include/smallk_amd.h is a C header: a C99 translation unit needs nothing else."""
    _build_c(tmp_path)


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    r = subprocess.run([str(_build_c(tmp_path))], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("iterations") == 3 and "k = 32" in r.stdout


@pytest.mark.gpu
def test_python_example_runs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "cluster_documents.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "documents per flat cluster" in r.stdout and "NMF: W (2000, 8)" in r.stdout


def test_pysmallk_clustering_classes_import_from_the_package():
    """pysmallk's Hierclust / Flatclust (smallk_lib.pyx:924-1420) stay importable where rounds 1-3 had them."""
    import smallk_amd
    from smallk_amd import Flatclust, Hierclust
    import smallk_amd.pyclust as pc
    assert pc.Hierclust is Hierclust and pc.Flatclust is Flatclust and smallk_amd.pyclust is pc
    for cls in (Hierclust, Flatclust):
        for name in ("load_matrix", "load_dictionary", "cluster", "get_top_indices", "get_assignments", "write_output", "finalize"):
            assert callable(getattr(cls, name)), (cls.__name__, name)
    with pytest.raises(AttributeError):
        smallk_amd.no_such_thing
