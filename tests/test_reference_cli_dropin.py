"""Drop-in proof with the reference's OWN caller: example/llz_resample (main.c + llz_parseopt.c + llz_wavfmt.c),
compiled unmodified in the build container (oracle/Makefile `cli`), once against the reference's libllzfilter sources
and once against libllzfilter_hip.so.  On the GPU box the relinked binary must write byte-identical WAV files.
(The binaries live in oracle/_ref/, are git-ignored and travel with the snapshot; absent -> skipped.)"""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "llz_resample_ref")
HIP = os.path.join(ROOT, "oracle", "_ref", "llz_resample_hip")

pytestmark = pytest.mark.gpu


def write_wav(path, pcm, rate):
    data = pcm.astype("<i2").tobytes()
    hdr = (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
           struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) + b"data" + struct.pack("<I", len(data)))
    with open(path, "wb") as f:
        f.write(hdr + data)


def synth_wav(path, seconds=0.5, rate=48000, seed=0):
    t = np.arange(int(seconds * rate))
    rng = np.random.default_rng(seed)
    x = 0.5 * 32767 * np.sin(2 * np.pi * 1000 * t / rate) + rng.integers(-2000, 2000, len(t))
    write_wav(path, np.clip(x, -32768, 32767).astype(np.int16), rate)


def run(exe, args, cwd):
    return subprocess.run([exe] + args, cwd=cwd, capture_output=True, text=True, timeout=300)


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIP)), reason="reference CLI binaries not built")
@pytest.mark.parametrize("up,down", [(147, 160), (160, 147), (1, 3), (2, 3), (3, 2)])
def test_reference_cli_relinked_is_byte_identical(tmp_path, up, down):
    src = os.path.join(str(tmp_path), "in.wav")
    synth_wav(src, seed=up * 7 + down)
    outs = {}
    for name, exe in (("ref", REF), ("hip", HIP)):
        out = os.path.join(str(tmp_path), name + ".wav")
        r = run(exe, ["-i", src, "-o", out, "-t", "2", "-u", str(up), "-d", str(down)], str(tmp_path))
        assert r.returncode == 0, (name, r.stdout[-300:], r.stderr[-300:])
        outs[name] = open(out, "rb").read()
    assert len(outs["ref"]) > 44 and outs["ref"] == outs["hip"]
    assert struct.unpack("<I", outs["hip"][24:28])[0] == 48000 * up // down          # header rate, e.g. 44100


@pytest.mark.skipif(not os.path.exists(HIP), reason="relinked CLI not built")
def test_relinked_cli_decimate_and_interp_do_not_crash(tmp_path, oracle):
    """-t 0 / -t 1 call llz_resample_filter_uninit on a decimate/interp handle (reference main.c:125): the CPU
    library crashes there for -t 0; this library accepts it.  Samples are checked against the oracle."""
    src = os.path.join(str(tmp_path), "in.wav")
    synth_wav(src, seconds=0.25, seed=5)
    pcm = np.frombuffer(open(src, "rb").read()[44:], dtype="<i2")
    for t, (L, M, mode) in {"0": (1, 3, 0), "1": (3, 1, 1)}.items():
        out = os.path.join(str(tmp_path), f"t{t}.wav")
        r = run(HIP, ["-i", src, "-o", out, "-t", t, "-u", str(L), "-d", str(M)], str(tmp_path))
        assert r.returncode == 0, (t, r.stdout[-300:], r.stderr[-300:])
        got = np.frombuffer(open(out, "rb").read()[44:], dtype="<i2")
        nin = oracle.rs_info(mode, L, M, 1.0, 1)["bytes_in"] // 2
        frames = -(-len(pcm) // nin)                                                 # the CLI zero-pads the last frame
        padded = np.zeros(frames * nin, dtype=np.int16)
        padded[:len(pcm)] = pcm
        ref = oracle.rs_stream(mode, L, M, 1.0, 1, padded)
        assert np.array_equal(got[:len(ref)], ref[:len(got)]) and len(got) >= len(ref) - nin * max(L, 1)
