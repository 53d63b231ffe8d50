"""ORACLE - TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never imported by the product package time-r1_amd/."""
