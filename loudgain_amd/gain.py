"""Caller-side arithmetic of loudgain's main loop: what
/root/reference/src/loudgain.c:323-379 does with a scan_result -- peak after
gain, clipping prediction, clipping prevention (-k / -K) -- and the `-O`
(--output-new) row format of loudgain.c:586-612.  Pure host fp64; it sits above
the scan boundary and never touches the GPU.  Pinned by the three tables the
reference's README shows (docs/images/test-{1,2,3}.csv.png; tests/test_gain_readme.py).
"""
import math


def gain_to_q78num(gain):
    """Opus R128 gain as Q7.8 (src/tag.cc:442-445): (int) round(gain * 256), C rounding."""
    x = gain * 256.0
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def apply_clip_logic(track_gain, track_peak, album_gain=0.0, album_peak=0.0, do_album=False,
                     no_clip=False, max_true_peak_level=-1.0):
    """loudgain.c:323-379.  Returns the (possibly corrected) gains and the flags
    the output rows show."""
    tpeak = 10.0 ** (max_true_peak_level / 20.0)  # track peak limit
    apeak = tpeak                                  # album peak limit
    tclip = aclip = False
    tgain = 10.0 ** (track_gain / 20.0) * track_peak  # track peak after gain
    tnew = tgain
    again = anew = 1.0
    if do_album:
        again = 10.0 ** (album_gain / 20.0) * album_peak
        anew = again
    will_clip = (tgain > tpeak) or (do_album and again > apeak)
    if will_clip and no_clip:
        if tgain > tpeak:
            tnew = min(tgain, tpeak)
            track_gain = track_gain - math.log10(tgain / tnew) * 20.0
            tclip = True
        if do_album and again > apeak:
            anew = min(again, apeak)
            album_gain = album_gain - math.log10(again / anew) * 20.0
            aclip = True
        will_clip = False
    return dict(track_gain=track_gain, album_gain=album_gain, will_clip=will_clip, tclip=tclip,
                aclip=aclip, tnew=tnew, anew=anew, again=again, apeak=apeak)


def _db(x):
    return 20.0 * math.log10(x) if x > 0 else -math.inf


def output_new_header():
    return ("File\tLoudness\tRange\tTrue_Peak\tTrue_Peak_dBTP\tReference\tWill_clip\tClip_prevent\t"
            "Gain\tNew_Peak\tNew_Peak_dBTP")


def output_new_rows(scan, clip, last, do_album, unit="dB"):
    """The `-O` lines of one file (loudgain.c:586-612); `scan` has scan_result's
    fields, `clip` is apply_clip_logic's answer."""
    rows = ["%s\t%.2f LUFS\t%.2f %s\t%.6f\t%.2f dBTP\t%.2f LUFS\t%s\t%s\t%.2f %s\t%.6f\t%.2f dBTP" % (
        scan["file"], scan["track_loudness"], scan["track_loudness_range"], unit, scan["track_peak"],
        _db(scan["track_peak"]), scan["loudness_reference"], "Y" if clip["will_clip"] else "N",
        "Y" if clip["tclip"] else "N", clip["track_gain"], unit, clip["tnew"], _db(clip["tnew"]))]
    if last and do_album:
        rows.append("%s\t%.2f LUFS\t%.2f %s\t%.6f\t%.2f dBTP\t%.2f LUFS\t%s\t%s\t%.2f %s\t%.6f\t%.2f dBTP" % (
            "Album", scan["album_loudness"], scan["album_loudness_range"], unit, scan["album_peak"],
            _db(scan["album_peak"]), scan["loudness_reference"],
            "Y" if (not clip["aclip"] and clip["again"] > clip["apeak"]) else "N",
            "Y" if clip["aclip"] else "N", clip["album_gain"], unit, clip["anew"], _db(clip["anew"])))
    return rows
