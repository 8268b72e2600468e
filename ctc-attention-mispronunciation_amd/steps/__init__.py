"""Host-side mirrors of the reference's `steps/` entry points that sit right after the hot path."""
