"""MI355X-native GAN-vocoder engine behind the ParallelWaveGAN Python surface."""
__version__ = "0.1.0"
