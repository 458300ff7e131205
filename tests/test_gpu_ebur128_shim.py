"""GPU: the libebur128-compatible entry points (include/loudscan_ebur128.h) driven the
way /root/reference/src/scan.c drives libebur128, against the oracle's restatement of
the same functions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _s16(frames, ch, rate, seed, gain=1.0):
    from loudgain_amd import synth
    f = synth.track_numpy(frames, ch, rate, seed=seed, step_s=1.5) * gain
    return np.clip(np.rint(f * 32768.0), -32768, 32767).astype(np.int16)


def _oracle_state(oracle, pcm16, rate):
    st = oracle.State(pcm16.shape[1], rate)
    st.add((pcm16.astype(np.float32) / 32768.0))   # add_frames_short scales by 1/32768 (exact in f32)
    return st


@pytest.mark.parametrize("rate,ch", [(48000, 2), (44100, 1), (96000, 2), (48000, 6)])
def test_state_like_scan_c(oracle, rate, ch):
    from loudgain_amd import ebur128
    pcm = _s16(int(rate * 13.7), ch, rate, seed=rate % 97 + ch)
    st = ebur128.State(ch, rate)                      # scan.c:203, all five modes
    assert st.channels == ch                          # scan.c:300 reads ->channels
    for a in range(0, pcm.shape[0], 1152):            # decoder-frame sized pieces, scan.c:448
        st.add_frames(pcm[a:a + 1152])
    ref = _oracle_state(oracle, pcm, rate)
    assert abs(st.loudness_global() - ref.loudness()) <= 1e-6      # scan.c:294
    assert abs(st.loudness_range() - ref.lra()) <= 1e-6            # scan.c:297
    for c in range(ch):                                            # scan.c:302-306
        assert abs(st.true_peak(c) - ref.true_peak(c)) <= 1e-4
        assert abs(st.sample_peak(c) - ref.sample_peak(c)) <= 1e-7
    with pytest.raises(ebur128.Ebur128Error) as e:
        st.true_peak(ch)
    assert e.value.code == ebur128.ERROR_INVALID_CHANNEL_INDEX
    st.close()


def test_multiple_like_scan_set_album_result(oracle):
    from loudgain_amd import ebur128
    specs = [(48000, 2, 9.0, 1, 1.0), (48000, 2, 7.5, 2, 0.05), (44100, 2, 11.0, 3, 0.7), (48000, 1, 4.0, 4, 1.0)]
    sts, refs = [], []
    for rate, ch, secs, seed, g in specs:
        pcm = _s16(int(rate * secs), ch, rate, seed, g)
        sts.append(ebur128.State(ch, rate).add_frames(pcm))
        refs.append(_oracle_state(oracle, pcm, rate))
    for s, r in zip(sts, refs):                       # per-track queries first, like loudgain does
        assert abs(s.loudness_global() - r.loudness()) <= 1e-6
    assert abs(ebur128.loudness_global_multiple(sts) - oracle.album_loudness(refs)) <= 1e-6   # scan.c:383
    assert abs(ebur128.loudness_range_multiple(sts) - oracle.album_lra(refs)) <= 1e-6         # scan.c:388
    # a subset is its own album
    assert abs(ebur128.loudness_global_multiple(sts[:2]) - oracle.album_loudness(refs[:2])) <= 1e-6


def test_modes_float_frames_and_streaming(oracle):
    from loudgain_amd import ebur128, synth
    rate, ch = 48000, 2
    f = synth.track_numpy(rate * 6, ch, rate, seed=9, step_s=1.0).astype(np.float32)
    st = ebur128.State(ch, rate, ebur128.MODE_I)       # integrated loudness only
    st.add_frames(f[: rate * 3])
    ref = oracle.State(ch, rate).add(f[: rate * 3])
    assert abs(st.loudness_global() - ref.loudness()) <= 1e-6
    st.add_frames(f[rate * 3:])                       # more frames after a query: results follow
    ref.add(f[rate * 3:])
    assert abs(st.loudness_global() - ref.loudness()) <= 1e-6
    for q in (st.loudness_range, lambda: st.true_peak(0), lambda: st.sample_peak(0)):
        with pytest.raises(ebur128.Ebur128Error) as e:
            q()
        assert e.value.code == ebur128.ERROR_INVALID_MODE
    with pytest.raises(ebur128.Ebur128Error):          # one sample type per state
        st.add_frames(np.zeros((4, ch), np.int16))
    # silence / too short: -inf like libebur128 (-HUGE_VAL), LRA 0
    z = ebur128.State(1, 44100).add_frames(np.zeros((1000, 1), np.int16))
    assert z.loudness_global() == -np.inf and z.loudness_range() == 0.0


def test_init_limits():
    from loudgain_amd import ebur128
    for ch, rate in [(0, 48000), (65, 48000), (2, 8), (2, 3000000)]:
        with pytest.raises(ebur128.Ebur128Error):
            ebur128.State(ch, rate)


def test_loudgain_main_sequence_is_one_plan(oracle):
    """/root/reference/src/loudgain.c:299-340 through scan.c: every file scanned (init + add_frames per decoded
    frame), THEN per file scan_get_track_result (loudness_global, loudness_range, true_peak per channel:
    scan.c:294-306) and scan_set_album_result (both _multiple calls over all states: scan.c:383-391) and finally
    scan_get_album_peak (true_peak of every channel of every state: scan.c:359-378).  One batched scan serves it."""
    from loudgain_amd import ebur128
    specs = [(48000, 2, 8.0, 11, 1.0), (48000, 2, 6.5, 12, 0.3), (44100, 1, 7.0, 13, 0.8), (48000, 6, 5.0, 14, 1.0),
             (96000, 2, 4.2, 15, 0.6)]
    sts, refs = [], []
    for rate, ch, secs, seed, g in specs:
        pcm = _s16(int(rate * secs), ch, rate, seed, g)
        st = ebur128.State(ch, rate)
        for a in range(0, pcm.shape[0], 4096):
            st.add_frames(pcm[a:a + 4096])
        sts.append(st)
        refs.append(_oracle_state(oracle, pcm, rate))
    before = ebur128.plan_count()
    for st, ref in zip(sts, refs):
        assert abs(st.loudness_global() - ref.loudness()) <= 1e-6
        assert abs(st.loudness_range() - ref.lra()) <= 1e-6
        for c in range(st.channels):
            assert abs(st.true_peak(c) - ref.true_peak(c)) <= 1e-4
        assert abs(ebur128.loudness_global_multiple(sts) - oracle.album_loudness(refs)) <= 1e-6
        assert abs(ebur128.loudness_range_multiple(sts) - oracle.album_lra(refs)) <= 1e-6
    peak = max(st.true_peak(c) for st in sts for c in range(st.channels))
    assert abs(peak - max(r.peak() for r in refs)) <= 1e-4
    assert ebur128.plan_count() - before == 1
    # frames arriving after the queries: the next query scans again (once), results follow
    extra = _s16(48000, 2, 48000, 99)
    sts[0].add_frames(extra)
    refs[0].add(extra.astype(np.float32) / 32768.0)
    assert abs(sts[0].loudness_global() - refs[0].loudness()) <= 1e-6
    assert abs(ebur128.loudness_global_multiple(sts) - oracle.album_loudness(refs)) <= 1e-6
    assert ebur128.plan_count() - before == 3     # the stale state alone, then the album over all five again
    for st in sts:
        st.close()
