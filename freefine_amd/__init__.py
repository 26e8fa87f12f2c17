"""freefine_amd: MI355X-native engine for the FreeFine DDIM-inversion + guided-denoising hot path."""
__version__ = "0.1.0"
