"""File formats either side of the weight-application path."""
