"""CPU: the host logic of the library batcher (folder clustering as bin/rgbpm2 does it,
album dealing, batch packing).  No GPU, no compute."""
import os


def _touch(path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "wb").close()


def test_cluster_like_rgbpm2(tmp_path):
    from loudgain_amd.batch import EXTENSIONS, album_tasks, cluster_folders
    root = str(tmp_path)
    for rel in ["a/01.wav", "a/02.WAV", "a/cover.jpg", "a/03.flac", "b/x.wav", "b/sub/y.wav",
                "various [compilations]/z.wav", "c/readme.txt"]:
        _touch(os.path.join(root, rel))
    cl, excluded = cluster_folders([root], extensions=EXTENSIONS)
    assert excluded == 1                                   # bin/rgbpm2:79,127-133
    tasks = album_tasks(cl)
    got = {(os.path.relpath(t["folder"], root), t["ext"]): [os.path.basename(f) for f in t["files"]] for t in tasks}
    assert got == {("a", ".wav"): ["01.wav", "02.WAV"],     # same folder + same type = one album
                   ("a", ".flac"): ["03.flac"],            # another type in the folder = another album
                   ("b", ".wav"): ["x.wav"], (os.path.join("b", "sub"), ".wav"): ["y.wav"]}
    cl2, _ = cluster_folders([root])                        # default: what this build can read
    assert all(set(v) == {".wav"} for v in cl2.values())


def test_deal_albums_balanced_and_complete():
    from loudgain_amd.batch import deal_albums
    sizes = [50, 10, 10, 10, 10, 10, 30, 30]
    parts = deal_albums(sizes, 3)
    assert sorted(sum(parts, [])) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert max(loads) <= 60 and min(loads) >= 50
    assert all(p == sorted(p) for p in parts)
    assert deal_albums([], 2) == [[], []]
    assert deal_albums([5], 4) == [[0], [], [], []]


def test_pack_batches():
    from loudgain_amd.batch import pack_batches
    assert pack_batches([4, 4, 4, 9, 1, 1], 8) == [[0, 1], [2], [3], [4, 5]]
    assert pack_batches([], 8) == []
    assert pack_batches([100], 8) == [[0]]                  # an oversized album still gets scanned
