"""CPU: the on-disk cache of run-time compiled code objects is only ever a directory nobody but this
user can write to, and a cache file is loaded only when it is what the library stored
(include/kabc.h, kabc_set_specialize; csrc/capi_plugin.hip rtc_cache_dir / cache_load).

Code objects found in the cache go to hipModuleLoadData and run on the GPU of the calling process,
and their file names are computable (a hash of the unit's text), so a directory another local user
could have prepared -- a pre-created $TMPDIR/kabc_rtc_cache_<uid>, a 0777 directory, a symbolic
link -- must be refused, not used."""
import ctypes as C
import json
import os
import struct
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, json, os, sys, time
sys.path.insert(0, %(root)r)
os.environ.setdefault("KABC_NO_TORCH_PRELOAD", "1")
import kissabc_jl_amd as k
lib = k._lib.load()
buf = C.create_string_buffer(4096)
n = lib.kabc_rtc_cache_dir(buf, 4096)
out = {"dir": buf.value.decode(), "len": n}
if %(compile)r:
    model = k.ApproxKernelizedPosterior(k.Factored(k.Normal(0, 5), k.Beta(%(a)r, 3.0)), k.costs.GaussDist([1.0, 0.5]), 0.1)
    t0 = time.time()
    try:
        out["handle"] = k.compile_model(model, families=1)
        out["error"] = ""
    except Exception as e:   # (no device here: the load fails after the compilation, as kabc.h says)
        out["error"] = str(e)
    out["seconds"] = time.time() - t0
print(json.dumps(out))
'''


def _child(env, compile=False, a=2.0):
    e = dict(os.environ)
    e.pop("KABC_RTC_CACHE_DIR", None)
    e.pop("KABC_SPECIALIZE", None)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "compile": compile, "a": a}], env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_explicit_directory_must_be_private(tmp_path):
    good = tmp_path / "good"
    good.mkdir(mode=0o700)
    assert _child({"KABC_RTC_CACHE_DIR": str(good)})["dir"] == str(good)
    # group- or world-writable: anybody could have planted a code object
    for mode in (0o777, 0o770, 0o702):
        bad = tmp_path / f"bad{mode:o}"
        bad.mkdir()
        os.chmod(bad, mode)
        assert _child({"KABC_RTC_CACHE_DIR": str(bad)}) == {"dir": "", "len": 0}
    # a symbolic link, even to a private directory
    link = tmp_path / "link"
    link.symlink_to(good)
    assert _child({"KABC_RTC_CACHE_DIR": str(link)})["dir"] == ""
    # not a directory
    f = tmp_path / "file"
    f.write_text("x")
    assert _child({"KABC_RTC_CACHE_DIR": str(f)})["dir"] == ""
    # a missing one is created private
    new = tmp_path / "new"
    assert _child({"KABC_RTC_CACHE_DIR": str(new)})["dir"] == str(new)
    assert (os.stat(new).st_mode & 0o777) == 0o700
    # "" / "0" switch the cache off
    assert _child({"KABC_RTC_CACHE_DIR": "0"})["dir"] == ""


def _relocated_library(tmp_path):
    """the library reached through a link in a private directory: <that directory>/rtc_cache is the
    first candidate of the default chain, and the test owns it"""
    import kissabc_jl_amd as k
    libdir = tmp_path / "lib"
    libdir.mkdir(mode=0o700)
    (libdir / "libkabc_hip.so").symlink_to(k.LIB_PATH)
    real = os.path.dirname(k.LIB_PATH)
    env = {"KABC_LIB": str(libdir / "libkabc_hip.so"),
           "KABC_RTC_INCLUDE": os.path.join(real, "..", "csrc") + ":" + os.path.join(real, "..", "..", "include"),
           "KABC_RTC_WORKER": os.path.join(real, "kabc_rtc_worker")}
    return libdir, env


def test_default_chain_skips_hostile_candidates(tmp_path):
    libdir, env = _relocated_library(tmp_path)
    uid = os.geteuid()
    # 1. next to the library, private: taken
    assert _child(env)["dir"] == str(libdir / "rtc_cache")
    # 2. next to the library but world-writable: skipped for $XDG_CACHE_HOME/kabc_rtc_cache
    os.chmod(libdir / "rtc_cache", 0o777)
    xdg = tmp_path / "xdg"
    xdg.mkdir(mode=0o700)
    env2 = dict(env, XDG_CACHE_HOME=str(xdg))
    assert _child(env2)["dir"] == str(xdg / "kabc_rtc_cache")
    # 3. that one hostile too: $TMPDIR/kabc_rtc_cache_<uid>
    os.chmod(xdg / "kabc_rtc_cache", 0o777)
    tmp = tmp_path / "tmp"
    tmp.mkdir(mode=0o700)
    env3 = dict(env2, TMPDIR=str(tmp))
    assert _child(env3)["dir"] == str(tmp / f"kabc_rtc_cache_{uid}")
    # 4. THE case of the round-5 review: somebody pre-created the predictable name with the wrong mode
    #    (before: accepted because it existed and was writable) -- refused, and nothing else is left
    os.chmod(tmp / f"kabc_rtc_cache_{uid}", 0o777)
    got = _child(env3, compile=True, a=2.125)
    assert got["dir"] == "" and got["len"] == 0
    # the model still compiles (no cache: every time) and nothing was written into the hostile places
    assert "cannot" not in got["error"].lower() or "hipModuleLoadData" in got["error"]
    for d in (libdir / "rtc_cache", xdg / "kabc_rtc_cache", tmp / f"kabc_rtc_cache_{uid}"):
        assert os.listdir(d) == []


def _fnv1a(data, h):
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def _parse_co(path):
    """the cache file of csrc/capi_plugin.hip cache_store: magic, key digest, code checksum, names, code"""
    raw = open(path, "rb").read()
    assert raw[:8] == b"KABCRTC2"
    digest, ck, n = struct.unpack_from("<QQI", raw, 8)
    off = 28
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", raw, off)
        off += 4 + ln
    (cs,) = struct.unpack_from("<Q", raw, off)
    code = raw[off + 8:off + 8 + cs]
    return {"digest": digest, "checksum": ck, "code_off": off + 8, "code": code, "size": len(raw),
            "ok": len(code) == cs and _fnv1a(code[:4096], 0) is not None}


def test_corrupted_and_foreign_cache_files_are_not_loaded(tmp_path):
    cache = tmp_path / "cache"
    cache.mkdir(mode=0o700)
    env = {"KABC_RTC_CACHE_DIR": str(cache), "KABC_SPECIALIZE": "1"}
    first = _child(env, compile=True, a=2.375)
    cos = [n for n in os.listdir(cache) if n.endswith(".co")]
    assert len(cos) == 1, os.listdir(cache)
    path = cache / cos[0]
    assert (os.stat(path).st_mode & 0o077) == 0           # private file
    good = open(path, "rb").read()
    info = _parse_co(path)
    # stored checksum = FNV-1a of the code bytes with the library's second basis
    assert _fnv1a(info["code"], 0x84222325cbf29ce4) == info["checksum"]
    # a cache hit leaves the file alone (a stored file is a NEW file: written beside, renamed over)
    ino = os.stat(path).st_ino
    _child(env, compile=True, a=2.375)
    assert os.stat(path).st_ino == ino and open(path, "rb").read() == good

    def run_and_expect_recompiled(label):
        before = os.stat(path).st_ino
        _child(env, compile=True, a=2.375)
        # not loaded: compiled again and stored again, intact, as a new file
        assert open(path, "rb").read() == good, label
        assert os.stat(path).st_ino != before, label

    # (a) flipped bytes in the code object
    bad = bytearray(good)
    for i in range(info["code_off"] + 100, info["code_off"] + 164):
        bad[i] ^= 0x5A
    open(path, "wb").write(bytes(bad))
    run_and_expect_recompiled("flipped code bytes")
    # (b) truncated
    open(path, "wb").write(good[:len(good) // 2])
    run_and_expect_recompiled("truncated")
    # (c) a valid file of ANOTHER unit under this name (wrong key digest)
    other = bytearray(good)
    struct.pack_into("<Q", other, 8, info["digest"] ^ 1)
    open(path, "wb").write(bytes(other))
    run_and_expect_recompiled("wrong digest")
    # (d) a symbolic link where the file should be
    target = tmp_path / "elsewhere.co"
    target.write_bytes(good)
    os.remove(path)
    os.symlink(target, path)
    _child(env, compile=True, a=2.375)
    assert not os.path.islink(path) and open(path, "rb").read() == good
    assert target.read_bytes() == good                    # nothing was written through the link


def test_set_specialize_overrides_the_environment(k, monkeypatch, tmp_path):
    """kabc_set_specialize: an embedding host forbids the worker process without an environment variable"""
    cache = tmp_path / "c"
    cache.mkdir(mode=0o700)
    monkeypatch.setenv("KABC_RTC_CACHE_DIR", str(cache))
    monkeypatch.delenv("KABC_SPECIALIZE", raising=False)
    lib = k._lib.load()
    out = (C.c_uint64 * 4)()
    lib.kabc_spec_counters(out)
    spawned0 = out[0]
    model = k.ApproxKernelizedPosterior(k.Factored(k.Normal(0, 5), k.Beta(2.625, 3.0)), k.costs.GaussDist([1.0, 0.5]), 0.1)
    cm = model.to_c()
    try:
        k.set_specialize("off")
        k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))
        lib.kabc_spec_counters(out)
        assert out[0] == spawned0 and os.listdir(cache) == []
        k.set_specialize("background")
        monkeypatch.setenv("KABC_SPECIALIZE", "0")            # the host's word beats the environment
        k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))
        lib.kabc_spec_counters(out)
        assert out[0] == spawned0 + 1
        with pytest.raises(k.KabcError):
            k._lib.check(lib.kabc_set_specialize(7))
    finally:
        k.set_specialize("env")
    # let the worker finish before the directory goes
    t0 = time.time()
    while time.time() - t0 < 300 and any(n.endswith((".lock", ".job")) for n in os.listdir(cache)):
        time.sleep(0.2)
