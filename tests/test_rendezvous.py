"""The file rendezvous of the multi-GPU host (include/iile_dist.h, iile_dist_rendezvous_file_token): ranks != 0 must only
ever take an id that belongs to THIS launch — a file left by an earlier run holds a dead RCCL id and would hang
ncclCommInitRank. No GPU needed: the reader side is plain file logic."""
import ctypes
import os
import struct
import threading
import time


def _record(token, fill):
    return b"IILEDIST" + struct.pack("<Q", token) + bytes([fill]) * 128


def _wait(binding, path, rank, token, timeout_s):
    lib = binding.dist_lib()
    buf = (ctypes.c_uint8 * 128)()
    rc = lib.iile_dist_rendezvous_file_token(str(path).encode(), rank, token, buf, timeout_s)
    return rc, bytes(buf), lib.iile_dist_last_error().decode()


def test_stale_file_without_token_is_rejected(binding, tmp_path):
    f = tmp_path / "rv"
    f.write_bytes(_record(0, 7))
    old = time.time() - 3600
    os.utime(f, (old, old))
    rc, _, err = _wait(binding, f, 1, 0, 1)
    assert rc != 0 and "older than this process" in err


def test_fresh_file_without_token_is_accepted(binding, tmp_path):
    f = tmp_path / "rv"
    out = {}
    t = threading.Thread(target=lambda: out.update(r=_wait(binding, f, 1, 0, 10)))
    t.start()
    time.sleep(0.3)  # the waiting rank started first; "rank 0" publishes afterwards (temporary file + rename)
    tmp = tmp_path / "rv.tmp"
    tmp.write_bytes(_record(0, 9))
    os.rename(tmp, f)
    t.join()
    rc, got, err = out["r"]
    assert rc == 0, err
    assert got == bytes([9]) * 128


def test_token_must_match(binding, tmp_path):
    f = tmp_path / "rv"
    f.write_bytes(_record(1234, 5))
    rc, _, err = _wait(binding, f, 1, 999, 1)
    assert rc != 0 and "token mismatch" in err
    rc, got, err = _wait(binding, f, 1, 1234, 1)
    assert rc == 0 and got == bytes([5]) * 128
    # a tokenless reader does not take a tokened file either (it belongs to some launcher's job)
    rc, _, err = _wait(binding, f, 2, 0, 1)
    assert rc != 0


def test_old_format_and_garbage_are_rejected(binding, tmp_path):
    f = tmp_path / "rv"
    f.write_bytes(bytes(128))  # round 2's bare 128-byte id
    rc, _, err = _wait(binding, f, 1, 0, 1)
    assert rc != 0 and "timed out" in err


def test_stale_then_republished(binding, tmp_path):
    """The race of ADVICE r02: a non-root rank starts while the previous run's file is still there, rank 0 replaces it later."""
    f = tmp_path / "rv"
    f.write_bytes(_record(0, 1))
    old = time.time() - 600
    os.utime(f, (old, old))
    out = {}
    t = threading.Thread(target=lambda: out.update(r=_wait(binding, f, 1, 0, 10)))
    t.start()
    time.sleep(0.4)
    os.remove(f)  # rank 0: remove, then publish
    tmp = tmp_path / "rv.tmp"
    tmp.write_bytes(_record(0, 2))
    os.rename(tmp, f)
    t.join()
    rc, got, err = out["r"]
    assert rc == 0, err
    assert got == bytes([2]) * 128


def test_file_published_before_this_rank_called_is_accepted(binding, tmp_path):
    """rank 0 published, and this rank reaches the call seconds later (HIP initialisation of eight processes
    starting together): the file was there at entry, so it is judged by its age against the PROCESS (library load), not against
    the call — round 3 compared with the reader's clock at call time and one second of slack, and timed out after 120 s."""
    f = tmp_path / "rv"
    f.write_bytes(_record(0, 3))
    time.sleep(2.2)  # "HIP initialisation"
    rc, got, err = _wait(binding, f, 1, 0, 2)
    assert rc == 0, err
    assert got == bytes([3]) * 128


def test_cli_job_token_is_decimal_and_checked(binding):
    """iile_pbrt --job: decimal, whole argument, > 0 ('0123489' in base 0 is octal and stops at the 8; 'abc' became 0 =
    token-less mode in round 3)."""
    import subprocess
    exe = os.path.join(os.path.dirname(binding.__file__), "lib", "iile_pbrt")
    for bad in ("abc", "0", "012x", "-5", ""):
        p = subprocess.run([exe, "nothing.pbrt", "--gpurank", "0/1", "--rendezvous", "/tmp/x", "--job", bad], capture_output=True, text=True)
        assert p.returncode == 1 and "--job wants a decimal number" in p.stderr, (bad, p.stderr)
    script = open(os.path.join(os.path.dirname(os.path.dirname(binding.__file__)), "tools", "multi_gpu_cmdline.sh")).read()
    assert "JOB=1$RANDOM$RANDOM" in script
