"""ctypes prototypes of the BoomerAMG entry points (include/hypre_amd_parcsr_ls.h)."""
PROTOTYPES = {}
