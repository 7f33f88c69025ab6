"""ORACLE (test infrastructure) -- import-only stub of `av`: the reference imports it at module load
(data_utils/audio.py, data_utils/utils.py); nothing on the encoder path calls it."""
