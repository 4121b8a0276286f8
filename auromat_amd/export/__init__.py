"""Export of mappings to files (reference auromat/export/): netCDF following CF-1.6 / NODC conventions."""
