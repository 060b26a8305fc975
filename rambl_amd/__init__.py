"""MI355X-native StrainCall hot path of homopolymer/RAMBL."""
__all__ = ["cli", "ingest", "samio", "synth", "capi", "stage5"]
