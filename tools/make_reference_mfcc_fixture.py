"""tests/golden/reference_corpus_mfcc.npz: a pin on the REAL reference's outputs.

flucoma-core ships `Resources/Data/flucoma_corpus_mfcc.json` -- "pre-analysed MFCCs of the FluCoMa demo audio files in small
slices" (Resources/Data/info.txt): for every slice of the concatenated demo files (`flucoma_corpus_slices.wav` holds the
slice points, `flucoma_corpus_files.json` the file order) the mean and standard deviation over the slice's frames of
BufMFCC's 13 coefficients (startCoeff 1, 40 bands, 20 Hz - 20 kHz, fft 1024 / hop 512, default padding), i.e. 26 numbers
per slice computed by a FluCoMa build.  tests/test_oracle.py recomputes all 299 slices of the first three files (the ones
whose position in the concatenation does not depend on files missing from this checkout) with both oracles, reading the
reference's files where they lie; this script cuts slices 1 - 4 (0.81 s of Constanzo-PreparedSnare-M.wav, prepared snare
recorded by Rodrigo Constanzo, Portugal, 2018 -- Resources/AudioFiles/-credits.txt: demonstration purposes only) and their
four rows of the JSON into a small fixture, so that the HIP path can be held against the same reference numbers on the
GPU box, where /root/reference does not exist.

    python tools/make_reference_mfcc_fixture.py        (needs /root/reference)
"""
import json, os, struct, wave
import numpy as np

R = "/root/reference/Resources/"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    w = wave.open(R + "AudioFiles/Constanzo-PreparedSnare-M.wav", "rb")
    assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 44100)
    pcm = np.frombuffer(w.readframes(w.getnframes()), "<i2")
    raw = open(R + "Data/flucoma_corpus_slices.wav", "rb").read()
    i = raw.find(b"data")
    n = struct.unpack("<I", raw[i + 4:i + 8])[0]
    points = np.frombuffer(raw[i + 8:i + 8 + n], "<f4").astype(np.int64)
    rows = json.load(open(R + "Data/flucoma_corpus_mfcc.json"))["data"]
    first, last = 1, 4
    a, b = int(points[first]), int(points[last + 1])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "reference_corpus_mfcc.npz"),
                        pcm16=pcm[a:b], offset=np.int64(a), points=points[first:last + 2] - a,
                        expected=np.array([rows["%d.000000" % k] for k in range(first, last + 1)], dtype=np.float64),
                        sample_rate=np.float64(44100.0),
                        source=np.array("flucoma-core Resources/AudioFiles/Constanzo-PreparedSnare-M.wav samples %d..%d; "
                                        "Resources/Data/flucoma_corpus_mfcc.json rows %d..%d" % (a, b, first, last)))
    print("wrote", b - a, "samples,", last - first + 1, "slices")


if __name__ == "__main__":
    main()
