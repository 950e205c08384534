"""nanopore_amd -- MI355X-native drop-in for the banded pair-HMM realignment hot path of
mitenjain/nanopore (the `cactus_realign` fan-out behind nanopore/mappers and nanopore/analyses).
See DESIGN.md for scope and INTEGRATION.md for how the reference binds to it."""
__version__ = "0.1.0"
