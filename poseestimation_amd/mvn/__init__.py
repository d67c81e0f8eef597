"""Drop-in mirror of the reference's `mvn` package for the triangulation path:
`mvn.utils.op`, `mvn.utils.multiview`, `mvn.utils.volumetric`, `mvn.models.triangulation`."""
