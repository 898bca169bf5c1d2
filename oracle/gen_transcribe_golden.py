"""Writes tests/golden/transcribe_ref.json: outputs of the UNMODIFIED reference ``olmoasr/transcribe.py`` (run through
oracle/ref_transcribe_harness.py in the build container) on (a) the 40 scripted-decode cases of
``ref_transcribe_harness.scripted_cases()`` (token level) and the 24 of ``scripted_cr_cases()`` (with a tokenizer: texts,
compression-ratio fallback, initial_prompt), the 36 of ``scripted_words_cases()`` (word_timestamps / hallucination_silence_threshold with a
scripted ``add_word_timestamps``) and (b) a real tiny model with the timestamp bonus, 41 s of the seeded generator's
audio.  ``python -m oracle.gen_transcribe_golden``.  TEST INFRASTRUCTURE ONLY; the fixture travels to the GPU box, the reference
does not."""
import json
import os

import numpy as np
import torch

from oracle import decode_oracle as do
from oracle import mel_oracle as me
from oracle import model_oracle as mo
from oracle import ref_transcribe_harness as H

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "transcribe_ref.json")


def model_case():
    """Same model/audio/bias as tests/test_gpu_decode_parity.py::test_transcribe_timestamp_driven_seek."""
    dims = mo.Dims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2)
    sd = mo.init_state_dict(dims, seed=21, train_vocab_rows=False)
    sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"] * 3.0
    pcm = torch.cat([mo.synthetic_sample(300 + i)[0] for i in range(2)])[: 41 * 16000]
    mel_padded = torch.from_numpy(me.log_mel_spectrogram(pcm.numpy().astype(np.float32) / 32768.0, padding=480000).astype(np.float32))
    bias = torch.zeros(51864)
    bias[do.TIMESTAMP_BEGIN:] = 25.0
    kw = dict(temperature=0.0, logprob_threshold=None, no_speech_threshold=0.6, sample_len=7)
    return sd, dims, mel_padded, bias, kw


def main():
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    out = {"scripted": [], "scripted_cr": [], "scripted_words": [], "model": None}
    for c in H.scripted_cases():
        ref = H.run_reference(H.scripted_decode(c["seed"]), H.index_mel(c["content_frames"]), **dict(c["kw"]))
        out["scripted"].append(H.comparable(ref))
    for c in H.scripted_cr_cases():  # with a tokenizer: compression-ratio fallback, segment / result text, initial_prompt
        ref = H.run_reference(H.scripted_decode_cr(c["seed"]), H.index_mel(c["content_frames"]), **dict(c["kw"]))
        out["scripted_cr"].append(H.comparable(ref, text=True))
    for c in H.scripted_words_cases():  # word_timestamps=True (+ hallucination_silence_threshold): scripted add_word_timestamps on both sides
        ref = H.run_reference(H.scripted_decode_cr(c["seed"]), H.index_mel(c["content_frames"]), add_word_timestamps=H.scripted_words(c["seed"]),
                              **dict(c["kw"]))
        out["scripted_words"].append(H.comparable(ref, text=True, words=True))
    sd, dims, mel_padded, bias, kw = model_case()
    out["model"] = H.comparable(H.run_reference(H.model_decode(sd, dims, bias), mel_padded, **kw))
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(OUT, os.path.getsize(OUT), "bytes;", len(out["model"]["segments"]), "model segments")


if __name__ == "__main__":
    main()
