for lib in "" ntold ntlate ntexact "" ntold ntlate ntexact; do
  echo "== lib ${lib:-default}"
  if [ -n "$lib" ]; then MMGCLIP_HIP_LIB=$PWD/tools/libmmg_ab_$lib.so python tools/nt_dgelu_tiles.py child; else python tools/nt_dgelu_tiles.py child; fi
done
