from ._entry import main

main()
