"""ORACLE (test infrastructure) -- import-only stub of ``paddleaudio`` (data_utils/featurizer/audio_featurizer.py:3)."""
