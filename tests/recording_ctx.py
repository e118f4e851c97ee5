"""Moved to rten_amd/recording.py (bench.py's control-flow test mode no longer imports from tests/); kept as an alias for the tests."""
from rten_amd.recording import RecordingCtx, RecordingModel  # noqa: F401
