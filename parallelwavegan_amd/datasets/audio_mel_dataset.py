"""(audio, mel) utterance pairs from a dump directory -- the dataset contract of the reference's
``AudioMelDataset`` (/root/reference/parallel_wavegan/datasets/audio_mel_dataset.py:18-192): ``dataset[i]`` is
``(audio[T] float32, mel[T', C] float32)`` (``(utt_id, audio, mel)`` with ``return_utt_id``), utterances whose mel is
not longer than ``mel_length_threshold`` frames (or audio than ``audio_length_threshold`` samples) are dropped.

On-disk formats (bin/train.py:1125-1142 of the reference): ``format: npy`` -> ``<utt>-wave.npy`` + ``<utt>-feats.npy``;
``format: hdf5`` -> one ``<utt>.h5`` with the datasets ``wave`` and ``feats`` (needs h5py).  Kaldi scp input is Kaldi
glue (out of scope).  The items feed either the host ``Collater`` through a DataLoader or, uploaded once,
the HBM-resident ``DeviceCollater`` (bin/train.py of this package).
"""
import fnmatch
import logging
import os

import numpy as np


def find_files(root_dir, query="*.npy", include_root_dir=True):
    """Sorted list of the files under ``root_dir`` (recursively) whose name matches ``query``."""
    out = []
    for root, _, names in os.walk(root_dir, followlinks=True):
        for name in fnmatch.filter(names, query):
            out.append(os.path.join(root, name))
    if not include_root_dir:
        out = [f.replace(root_dir + "/", "") for f in out]
    return sorted(out)


def _loaders(fmt):
    if fmt == "npy":
        return "*-wave.npy", "*-feats.npy", np.load, np.load
    if fmt == "hdf5":
        from ..utils.utils import read_hdf5

        return "*.h5", "*.h5", (lambda p: read_hdf5(p, "wave")), (lambda p: read_hdf5(p, "feats"))
    raise ValueError(f"support only hdf5 or npy format (got {fmt!r})")


class AudioMelDataset(object):
    """torch.utils.data.Dataset-compatible (``__len__`` / ``__getitem__``) audio + mel pairs."""

    def __init__(self, root_dir, audio_query=None, audio_load_fn=None, mel_query=None, mel_load_fn=None,
                 audio_length_threshold=None, mel_length_threshold=None, return_utt_id=False, allow_cache=False,
                 format="npy"):
        dq_a, dq_m, dl_a, dl_m = _loaders(format) if None in (audio_query, audio_load_fn, mel_query, mel_load_fn) \
            else (None,) * 4
        audio_query, mel_query = audio_query or dq_a, mel_query or dq_m
        self.audio_load_fn, self.mel_load_fn = audio_load_fn or dl_a, mel_load_fn or dl_m
        audio_files, mel_files = find_files(root_dir, audio_query), find_files(root_dir, mel_query)
        assert len(audio_files) != 0, f"Not found any audio files in ${root_dir}."
        assert len(audio_files) == len(mel_files), \
            f"Number of audio and mel files are different ({len(audio_files)} vs {len(mel_files)})."
        keep = list(range(len(audio_files)))
        if audio_length_threshold is not None:
            keep = [i for i in keep if self.audio_load_fn(audio_files[i]).shape[0] > audio_length_threshold]
        if mel_length_threshold is not None:
            keep = [i for i in keep if self.mel_load_fn(mel_files[i]).shape[0] > mel_length_threshold]
        if len(keep) != len(audio_files):
            logging.warning(f"Some files are filtered by the length thresholds ({len(audio_files)} -> {len(keep)}).")
        assert keep, f"every utterance of {root_dir} is shorter than the thresholds"
        self.audio_files = [audio_files[i] for i in keep]
        self.mel_files = [mel_files[i] for i in keep]
        self.return_utt_id = return_utt_id
        strip = "-wave" if ".npy" in audio_query else ""
        self.utt_ids = [os.path.splitext(os.path.basename(f))[0].replace(strip, "") for f in self.audio_files]
        self.allow_cache = allow_cache
        self._cache = [None] * len(self.audio_files) if allow_cache else None

    def __len__(self):
        return len(self.audio_files)

    def __getitem__(self, idx):
        if self._cache is not None and self._cache[idx] is not None:
            return self._cache[idx]
        audio = np.asarray(self.audio_load_fn(self.audio_files[idx]), dtype=np.float32).reshape(-1)
        mel = np.asarray(self.mel_load_fn(self.mel_files[idx]), dtype=np.float32)
        item = (self.utt_ids[idx], audio, mel) if self.return_utt_id else (audio, mel)
        if self._cache is not None:
            self._cache[idx] = item
        return item
