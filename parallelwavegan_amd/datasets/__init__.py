from .audio_mel_dataset import AudioMelDataset, find_files  # noqa: F401
