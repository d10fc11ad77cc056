"""--bamFiles in the window loop, without a GPU: `dindel_gpu --prepareOnly` with DINDEL_DUMP_READS writes what every
window would hand to the likelihood step.  With several pools the order of the read buffer depends on the windows walked before
(DInDel.cpp:976-1003), so a worker that starts a batch replays the read selection of the batch's look-back first: the dumps of a run cut
into batches of 3 windows on 4 workers must equal those of one batch on one worker (the window-by-window history), reads in the same
order; a one-line --bamFiles list must equal --bamFile."""
import os
import subprocess

import numpy as np
import pytest

from tests import _bamwriter as bw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "dindel_tgi_amd", "host", "dindel_gpu")


def _env():
    env = dict(os.environ)
    try:
        import torch
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    except ImportError:
        pass
    return env


def _scene(tmp_path, n_ref=90000, n_reads=25000):
    rng = np.random.default_rng(11)
    recs = [[], []]
    for k in range(n_reads):
        pos = int(rng.integers(100, n_ref - 300))
        L = int(rng.integers(50, 120))
        seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, L))
        # few distinct mapping qualities: long ties, whose order is the buffer's
        recs[int(rng.integers(0, 2))].append(dict(qname="r%d" % k, flag=int(rng.choice([0, 16])), pos=pos, mapq=int(rng.choice([30, 40, 60])), cigar="%dM" % L, seq=seq,
                                                  qual=[30] * L, mtid=-1, mpos=-1, isize=0, tags={}))
    paths = []
    for k in range(2):
        recs[k].sort(key=lambda r: r["pos"])
        paths.append(str(tmp_path / ("pool%d.bam" % k)))
        bw.write_bam(paths[-1], "@SQ\tSN:20\tLN:%d\n" % n_ref, [("20", n_ref)], [(0, r) for r in recs[k]])
    windows, left = [], 3000
    while left < n_ref - 3000:
        windows.append((left, left + 120))
        left += int(rng.choice([60, 200, 400, 400, 900, 2600, 7000]))          # 7000: a gap wider than the buffer's span
    vf, hf = str(tmp_path / "windows.txt"), str(tmp_path / "haps.txt")
    with open(vf, "w") as f:
        for left, right in windows:
            f.write("20 %d %d %d,+A\n" % (left, right, left + 60))
    with open(hf, "w") as f:
        for i, (left, right) in enumerate(windows):
            f.write("W %d %d %d\nH %s\nH %s\n" % (i + 1, left, right, "A" * 121, "A" * 60 + "C" + "A" * 61))
    lst = str(tmp_path / "bams.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    one = str(tmp_path / "one.txt")
    open(one, "w").write(paths[0] + "  ignored words\n\n")
    return dict(paths=paths, list=lst, one=one, vf=vf, hf=hf, n=len(windows))


def _dumps(tmp_path, tag, scene, bam_args, extra):
    d = tmp_path / tag
    d.mkdir()
    env = _env()
    env["DINDEL_DUMP_READS"] = str(d / "w")
    subprocess.check_call([DRIVER] + bam_args + ["--varFile", scene["vf"], "--hapFile", scene["hf"], "--outputFile", str(d / "out"), "--prepareOnly", "--quiet"] + extra, env=env)
    return [open(str(d / ("w.%d" % (i + 1)))).read() for i in range(scene["n"])]


def test_pooled_batches_replay_the_buffer_history(tmp_path):
    if not os.path.exists(DRIVER):
        pytest.skip("dindel_gpu not built")
    s = _scene(tmp_path)
    serial = _dumps(tmp_path, "serial", s, ["--bamFiles", s["list"]], ["--batchWindows", "100000", "--prepareThreads", "1"])
    assert sum(1 for t in serial if t.count("\n") >= 20) > s["n"] // 2
    pools_seen = {line.split()[1] for t in serial for line in t.split("\n") if line}
    assert pools_seen == {"0", "1"}
    for batch, threads in ((3, 4), (1, 3), (17, 2)):
        cut = _dumps(tmp_path, "cut%d" % batch, s, ["--bamFiles", s["list"]], ["--batchWindows", str(batch), "--prepareThreads", str(threads)])
        assert cut == serial, "batches of %d windows" % batch
    # without the replay the order inside ties differs somewhere (the test has teeth): every batch then starts with an empty buffer
    bare = _dumps(tmp_path, "bare", s, ["--bamFiles", s["list"]], ["--batchWindows", "3", "--prepareThreads", "2", "--noLookBack"])
    assert bare != serial and [sorted(t.split("\n")) for t in bare] == [sorted(t.split("\n")) for t in serial]      # same reads, another order
    a = _dumps(tmp_path, "one_a", s, ["--bamFiles", s["one"]], ["--batchWindows", "5", "--prepareThreads", "2"])
    b = _dumps(tmp_path, "one_b", s, ["--bamFile", s["paths"][0]], ["--batchWindows", "64", "--prepareThreads", "1"])
    assert a == b


def test_reference_exit_paths_end_the_run(tmp_path):
    """Where the reference calls exit() inside getReads (inconsistent mate positions, DInDel.cpp:1134-1137; an unmapped read with two mapped
    mates, :1180-1183) the window loop must stop with a non-zero status — not write a skipped line and carry on."""
    if not os.path.exists(DRIVER):
        pytest.skip("dindel_gpu not built")
    n_ref = 20000
    mk = lambda q, pos, flag, mpos, L=60: dict(qname=q, flag=flag, pos=pos, mapq=60, cigar="%dM" % L, seq="ACGT" * (L // 4), qual=[30] * L, mtid=0, mpos=mpos, isize=0, tags={})
    recs = [mk("f%d" % i, 5000 + 3 * i, 0, -1) for i in range(30)]
    for r in recs:
        r["mtid"] = -1
    recs += [mk("pair", 5010, 1 + 64, 5100), mk("pair", 5090, 1 + 128, 5010)]          # first mate says its mate is at 5100; it is at 5090
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "bad.bam")
    bw.write_bam(bam, "@SQ\tSN:20\tLN:%d\n" % n_ref, [("20", n_ref)], [(0, r) for r in recs])
    vf, hf = str(tmp_path / "w.txt"), str(tmp_path / "h.txt")
    open(vf, "w").write("20 5000 5120 5060,+A\n20 9000 9120 9060,+A\n")
    open(hf, "w").write("W 1 5000 5120\nH %s\nW 2 9000 9120\nH %s\n" % ("A" * 121, "C" * 121))
    r = subprocess.run([DRIVER, "--bamFile", bam, "--varFile", vf, "--hapFile", hf, "--outputFile", str(tmp_path / "o"), "--prepareOnly", "--quiet"],
                       env=_env(), capture_output=True, text=True)
    assert r.returncode == 1 and "matepos inconsistency!" in r.stderr, (r.returncode, r.stderr)


def test_late_skip_re_prepares_the_windows_behind_it(tmp_path):
    """A window skipped AFTER read selection (the likelihood or genotyping step threw: "hapSize error.", DInDel.cpp:1369-1408) empties the
    reference's read buffer for the window behind it (:1404-1405); with several pools that changes the ORDER of the buffer, hence of the reads
    inside a mapping-quality tie.  The prepare workers run ahead of that knowledge; the writer re-prepares the windows in reach.  Truth =
    the same run with the late skips announced to the prepare stage (--lateSkipsKnown: buffers reset behind them, window by window on one
    worker); the pipeline cut into batches any way must hand the same reads in the same order to every window and write the same bytes."""
    if not os.path.exists(DRIVER):
        pytest.skip("dindel_gpu not built")
    s = _scene(tmp_path)
    inject = "4,9,10,23,%d,%d" % (s["n"] // 2, s["n"])                      # single ones, two in a row, the last window of the file
    late = ["--injectLateSkip", inject]
    plain = _dumps(tmp_path, "plain", s, ["--bamFiles", s["list"]], ["--batchWindows", "100000", "--prepareThreads", "1"])
    truth = _dumps(tmp_path, "truth", s, ["--bamFiles", s["list"]], late + ["--lateSkipsKnown", "--batchWindows", "100000", "--prepareThreads", "1"])
    glf_truth = open(str(tmp_path / "truth" / "out.glf.txt")).read()
    assert glf_truth.count("error_hapSize_error.") == 6
    # the reset matters: behind the injected windows some window gets the same reads in another order
    reordered = [i for i in range(s["n"]) if plain[i] != truth[i]]
    assert reordered and all(sorted(plain[i].split("\n")) == sorted(truth[i].split("\n")) for i in reordered)
    # a second, independent truth: the writer redoing every window one after the other (--windowByWindow: the reference's loop as it stands)
    wbw = _dumps(tmp_path, "wbw", s, ["--bamFiles", s["list"]], late + ["--windowByWindow", "--batchWindows", "7", "--prepareThreads", "2"])
    assert wbw == truth and open(str(tmp_path / "wbw" / "out.glf.txt")).read() == glf_truth
    for batch, threads in ((100000, 1), (3, 4), (1, 3), (17, 2)):
        tag = "late%d" % batch
        got = _dumps(tmp_path, tag, s, ["--bamFiles", s["list"]], late + ["--batchWindows", str(batch), "--prepareThreads", str(threads)])
        assert got == truth, "batches of %d windows" % batch
        assert open(str(tmp_path / tag / "out.glf.txt")).read() == glf_truth
    # one pool: a reset does not change what a window sees, and nothing is re-prepared
    one = _dumps(tmp_path, "one_late", s, ["--bamFile", s["paths"][0]], late + ["--batchWindows", "5", "--prepareThreads", "3"])
    one_plain = _dumps(tmp_path, "one_plain", s, ["--bamFile", s["paths"][0]], ["--batchWindows", "5", "--prepareThreads", "3"])
    assert one == one_plain
