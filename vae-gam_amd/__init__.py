"""vae-gam_amd: MI355X-native VAE-GAM train step (drop-in for dannyfa/VAE-GAM's hot path).

The directory name follows the project layout contract; import it as `vae_gam_amd`
(see the loader module `vae_gam_amd.py` at the repository root).
"""
__version__ = '0.1.0'
