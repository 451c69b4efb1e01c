set -x
bash tools/profile.sh cohort_h64 r06c
bash tools/profile.sh cohort_h128 r06c
bash tools/profile.sh cohort_h17 r06c
