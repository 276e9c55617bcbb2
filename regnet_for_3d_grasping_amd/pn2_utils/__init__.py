"""Host-side mirror of the reference's ``multi_model/utils/pn2_utils`` operator API, bound to
the MI355X kernels.  Same module names (``function``, ``modules``, ``nn``), same op / class
names and call signatures, so model code written against the reference imports unchanged."""
