"""Link-level checks of the drop-in boundary (tests/link): programs compiled against the C header and linked with
-lstenos, run on the GPU box.  The binaries are built by tests/link/Makefile (__graft_entry__.build()) into build/."""
import os
import subprocess

import pytest

from _libs import ROOT

pytestmark = pytest.mark.gpu


def _binary(name):
    path = os.path.join(ROOT, "build", name)
    if not os.path.exists(path):
        pytest.skip(f"build/{name} was not built (tests/link/Makefile)")
    return path


def test_c_program_against_the_header_and_the_library():
    """include/stenos.h is valid C99, every symbol a C caller uses resolves in libstenos.so, and the calls work."""
    p = subprocess.run([_binary("abi_link")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ABI_LINK_OK" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-500:])


def test_the_references_own_round_trip_test_runs_against_this_library():
    """The reference's tests/tests_comp_decomp.cpp (bytesoftype 1..15 x same / sorted / random x levels 0..5 x threads 1..8 x
    shrinking dst_size, tests_comp_decomp.cpp:93-211), compiled with the reference's own header and linked with this
    library.  The whole matrix takes hours; it aborts at the first failure (STENOS_ABORT), so a bounded run that is still
    going -- or has finished -- without an abort is a pass.  Two and a half minutes here (about 25 000 round trips, all of the "same" distribution: the cells below reach the others); a 960 s run
    (171 640 round trips, bytesoftype 1..11, every level) is kept in profiles/r03_ref_tests_long.json."""
    import tempfile

    with tempfile.TemporaryFile() as log:  # (a pipe would fill up: the test prints a line per round trip)
        try:
            rc = subprocess.run([_binary("ref_tests_comp_decomp")], stdout=log, stderr=subprocess.STDOUT, timeout=150).returncode
        except subprocess.TimeoutExpired:
            rc = None
        log.seek(0)
        out = log.read().decode(errors="replace")
    assert "Test error" not in out, out[-800:]
    assert rc in (None, 0), (rc, out[-800:])
    assert out.count("done") > 10000, "the test did not get far enough: " + out[-500:]


@pytest.mark.parametrize("distribution,bytesoftype", [("sorted", 4), ("random", 2), ("random", 8), ("sorted", 7), ("random", 1), ("same", 12)])
def test_the_references_round_trip_test_by_distribution(distribution, bytesoftype):
    """The same test one (distribution, bytesoftype) cell at a time (tests/link/ref_tests_shard_main.cpp includes the
    reference's source from where it lies and calls its TestDistribution<K, K + 1>::apply): the reference's own driver
    needs hours to leave "same", so its bounded runs never reached "sorted" and "random".  A bounded run per cell; a cell
    that is still going without a failed check is a pass."""
    import tempfile

    with tempfile.TemporaryFile() as log:
        try:
            rc = subprocess.run([_binary("ref_tests_shard"), distribution, str(bytesoftype)], stdout=log, stderr=subprocess.STDOUT, timeout=25).returncode
        except subprocess.TimeoutExpired:
            rc = None
        log.seek(0)
        out = log.read().decode(errors="replace")
    assert "Test error" not in out, out[-800:]
    assert rc in (None, 0), (rc, out[-800:])
    assert out.count("done") > 500, "the cell did not get far enough: " + out[-500:]


def test_the_references_cvector_test_links_and_runs_against_this_library():
    """stenos/cvector.hpp (header only) drives stenos_private_compress_block / _decompress_block / _block_size ... one
    superblock per call (cvector.hpp:1383-1416).  Its own test (tests/test_cvector.cpp:75-740), compiled where it lies and
    linked with this library, must not fail a check; it takes far longer than a test slot here, because every access to
    a compressed bucket is a device call of ~0.2 ms (not a use this path is for), so a bounded run that is
    still going without a failed check is a pass."""
    import tempfile

    with tempfile.TemporaryFile() as log:
        try:
            rc = subprocess.run(["stdbuf", "-oL", "-eL", _binary("ref_test_cvector")], stdout=log, stderr=subprocess.STDOUT, timeout=60).returncode
        except subprocess.TimeoutExpired:
            rc = None
        log.seek(0)
        out = log.read().decode(errors="replace")
    assert rc in (None, 0), (rc, out[-800:])
    assert "rror" not in out and "failed" not in out.lower(), out[-800:]
