"""MI355X-native sparse solve path for ocean-tracer Jacobian-preconditioner systems."""
__version__ = "0.1.0"
