"""Generic helpers on the resampling path (mirror of the reference's ``auromat.util.histogram``)."""
